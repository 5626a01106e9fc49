"""Host-side mirror of the reference's replication / identity masks (crates/etl/src/schema.rs:30-129, 220-227, 380-441,
686-721; crates/etl/src/postgres/codec/event.rs:96-179): what a caller computes before it installs a Ready schema with
`Decoder.table_ready` (the table-copy path, crates/etl/src/replication/table_sync/mod.rs:280-300) and what a destination asks a
`ReplicatedTableSchema` about its row identity. Relation messages that arrive IN the stream are handled inside the library
(etl_amd/csrc/host.cpp handle_relation) with the same rules.

A stored column is (name, type_oid, nullable, primary_key_ordinal_or_0[, attnum]); attnum defaults to the 1-based position."""


class UnknownReplicatedColumns(Exception):
    """SchemaError::UnknownReplicatedColumns (schema.rs:30-61): the relation names columns the stored schema does not have."""

    def __init__(self, columns):
        self.columns = list(columns)
        super().__init__("Replication stream contains columns missing from the stored table schema: " + ", ".join(self.columns))


def _name(col):
    return col[0]


def _is_pk(col):
    return bool(col[3])


def _attnum(col, index):
    return col[4] if len(col) > 4 else index + 1


def replication_mask_try_build(columns, replicated_names):
    """ReplicationMask::try_build (schema.rs:99-129): 1 per stored column whose name the relation carries; every name must exist."""
    have = {_name(c) for c in columns}
    unknown = [n for n in replicated_names if n not in have]
    if unknown:
        raise UnknownReplicatedColumns(unknown)
    names = set(replicated_names)
    return [1 if _name(c) in names else 0 for c in columns]


def replication_mask_all(columns):
    """ReplicationMask::all."""
    return [1] * len(columns)


def replication_mask_build_or_all(columns, replicated_names):
    """ReplicationMask::build_or_all: fall back to every column when the names do not match the stored schema."""
    try:
        return replication_mask_try_build(columns, replicated_names)
    except UnknownReplicatedColumns:
        return replication_mask_all(columns)


def identity_mask_from_metadata(columns, replication_mask, primary_key_attnums, relreplident, replica_identity_index_attnums):
    """IdentityMessage::build_identity_mask (codec/event.rs:124-179): FULL = every replicated column, DEFAULT = the primary key,
    USING INDEX = the replica-identity index, NOTHING = empty — each intersected with the replication mask."""
    if relreplident == "f":
        return list(replication_mask)
    if relreplident == "n":
        return [0] * len(columns)
    if relreplident == "d":
        att = set(primary_key_attnums)
    elif relreplident == "i":
        att = set(replica_identity_index_attnums)
    else:
        raise ValueError(f"Invalid replica identity metadata: unsupported replica identity mode '{relreplident}'")
    return [1 if r == 1 and _attnum(c, i) in att else 0 for i, (c, r) in enumerate(zip(columns, replication_mask))]


def identity_mask_default(columns, replication_mask):
    """ReplicatedTableSchema::from_mask (schema.rs:446-460): the replicated primary-key columns."""
    return [1 if r == 1 and _is_pk(c) else 0 for c, r in zip(columns, replication_mask)]


def infer_identity_type(columns, replication_mask, identity_mask):
    """ReplicatedTableSchema::infer_identity_type (schema.rs:686-721)."""
    has_identity = False
    matches_pk = matches_full = True
    for c, r, i in zip(columns, replication_mask, identity_mask):
        has_identity |= i == 1
        if i != (1 if r == 1 and _is_pk(c) else 0):
            matches_pk = False
        if i != r:
            matches_full = False
    if not has_identity:
        return "Missing"
    if matches_pk:
        return "PrimaryKey"
    if matches_full:
        return "Full"
    return "AlternativeKey"


def unreplicated_primary_key_columns(columns, replication_mask):
    """ReplicatedTableSchema::unreplicated_primary_key_column_schemas."""
    return [_name(c) for c, r in zip(columns, replication_mask) if _is_pk(c) and r != 1]
