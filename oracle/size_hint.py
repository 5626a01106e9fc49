"""TEST INFRASTRUCTURE — CPU restatement of the reference's size hints, for the parity test of etlg_batch_size_hints
(etl_amd/csrc/columns.hip). Never imported by the product path.

PARITY UNPINNED: the reference has no test that states a size hint as a number, and the hints are sums of Rust
`size_of::<T>()` values that only a Rust build can supply — so this file restates the FORMULA (what is added for which
event / row / cell) and takes the sizes as a model, like the device does.

Follows: Event::size_hint crates/etl/src/event.rs:295-320 | TableRow::new + estimate_table_row_allocated_bytes
crates/etl/src/data/table_row.rs:28-32, 248-259 | estimate_cell_allocated_bytes :276-299 | numeric :302-307 |
Vec capacities: full rows codec/event.rs:567 (column_count), key rows :800 / :831 (identity_column_count), strings
codec/text.rs:151 (to_owned: len), bytea codec/hex.rs:21 (len / 2 of the hex digits = decoded length), numeric digits
etl-postgres/src/numeric.rs:444 (retained groups = ndigits).

Works on the event dicts of etl_amd.view.HostBatch.materialize()."""

INCOMPLETE = 1 << 63

# any consistent set of sizes will do for the test; these are what rustc 1.8x gives on x86_64 for the reference's types
MODEL = dict(begin_event=40, commit_event=48, insert_event=88, update_event=160, delete_event=104, truncate_event=64,
             relation_event=48, replicated_table_schema=56, table_row=32, cell=32)


def cell_bytes(cell):
    """(heap bytes, incomplete)"""
    k = cell[0]
    if k == "String":
        return len(cell[1]), False
    if k == "Bytes":
        return len(cell[1]), False
    if k == "Numeric":
        return (2 * len(cell[5]) if cell[1] == 0 else 0), False
    if k == "Missing":
        return 0, True
    if k == "Deferred":
        return 0, None     # depends on the column class: decided by the caller
    return 0, False


def row_bytes(cells, classes, capacity, model, text_form):
    total, inc = model["table_row"] + capacity * model["cell"], False
    for c, cls in zip(cells, classes):
        b, i = cell_bytes(c)
        if i is None:
            i = cls in text_form
        if c[0] == "String" and cls in text_form and cls != 8:
            b, i = 0, True        # a json / array column that came back as text: the host sizes the parsed value
        total += b
        inc = inc or i
    return total, inc


def event_hint(e, slots, model, text_form=(8, 16, 17)):
    k = e["kind"]
    if k == "B":
        return model["begin_event"]
    if k == "C":
        return model["commit_event"]
    if k == "R":
        return model["relation_event"]
    if k == "T":
        return model["truncate_event"] + len(e["tables"]) * model["replicated_table_schema"]
    slot = slots[e["schema_slot"]]
    full = [c.type_class for c in slot.cols]
    key = [c.type_class for c in slot.key_cols()]
    total = model[{"I": "insert_event", "U": "update_event", "D": "delete_event"}[k]]
    inc = False
    if k != "I" and e["old_kind"] != "None":
        iskey = e["old_kind"] == "Key"
        b, i = row_bytes(e["old_row"], key if iskey else full, len(key) if iskey else len(full), model, text_form)
        total, inc = total + b, inc or i
    if k != "D":
        b, i = row_bytes(e["row"], full, len(full), model, text_form)
        total, inc = total + b, inc or i or bool(e.get("partial"))
    return total | (INCOMPLETE if inc else 0)
