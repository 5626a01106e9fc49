// Deterministic synthetic pgoutput (proto v1, text tuples) stream generator,
// CopyData-framed exactly as on the socket. SplitMix64-seeded; used by the
// tests and bench.py for the workloads of SURVEY.md §8(d) (cfg 1-5).
//
// This is a measurement tool, not part of the decode path: it only WRITES
// the wire format (PostgreSQL "Logical Replication Message Formats", the
// same layouts the reference's test encoders produce at
// crates/etl/src/postgres/codec/event.rs:1076-1172).
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

namespace {

struct Rng {
  uint64_t s;
  uint64_t next() {
    uint64_t z = (s += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
  }
  uint64_t below(uint64_t n) { return n ? next() % n : 0; }
  uint64_t range(uint64_t lo, uint64_t hi) { return lo + below(hi - lo + 1); }  // inclusive
  bool pct(unsigned p) { return below(100) < p; }
};

enum ColKind : int32_t {
  CK_INT4_10D = 0,  // uniform [1e9, 2147483647]: always 10 digits
  CK_INT8 = 1, CK_INT4 = 2, CK_INT2 = 3, CK_BOOL = 4, CK_NUMERIC = 5, CK_TEXT = 6,
  CK_TIMESTAMPTZ = 7, CK_UUID = 8, CK_INT8_SEQ = 9
};

struct Buf {
  uint8_t* p; size_t cap; size_t n = 0; bool overflow = false;
  void put(const void* d, size_t k) { if (n + k > cap) { overflow = true; return; } memcpy(p + n, d, k); n += k; }
  void u8(uint8_t v) { put(&v, 1); }
  void be16(uint16_t v) { uint8_t b[2] = {(uint8_t)(v >> 8), (uint8_t)v}; put(b, 2); }
  void be32(uint32_t v) { uint8_t b[4] = {(uint8_t)(v >> 24), (uint8_t)(v >> 16), (uint8_t)(v >> 8), (uint8_t)v}; put(b, 4); }
  void be64(uint64_t v) { be32((uint32_t)(v >> 32)); be32((uint32_t)v); }
  void cstr(const char* s) { put(s, strlen(s) + 1); }
};

}  // namespace

extern "C" {

struct synth_col {
  int32_t kind;       // ColKind
  uint32_t type_oid;  // what the stored schema / Relation message says
  uint8_t nullable, pk, null_pct, utf8_pct;
  uint32_t min_len, max_len;  // CK_TEXT
  char name[32];
};

struct synth_table {
  uint32_t rel_id;
  uint32_t ncols;
  synth_col cols[32];
  char name[32];
};

struct synth_cfg {
  uint64_t seed;
  uint64_t start_lsn;
  uint32_t ntables;
  synth_table tables[4];
  uint32_t rows_per_txn;
  uint32_t pct_insert, pct_update, pct_delete;  // sums to 100
  uint32_t upd_pct_key, upd_pct_toast;          // of updates: K old tuple / 'u' on a text column
  uint32_t emit_relations;  // 1: R for every table inside the first txn that touches it
  uint32_t emit_origin;     // 1: one 'O' right after the first Begin
  uint32_t ddl_every_txns;  // 0: never; else every N txns one table gets M(ddl) -> R
  uint32_t type_msg_pct;    // % of R messages preceded by a 'Y'
  uint32_t keepalive_every; // 0: never; else a 'k' frame every N frames
};

struct synth_state {  // resumable across batches
  Rng rng;
  uint64_t lsn;
  uint64_t txn_no;
  uint64_t next_id;
  uint32_t frames_since_ka;
  uint8_t rel_sent[4];
  uint8_t origin_sent;
  uint32_t ddl_round;
  synth_table tables[4];  // current (post-DDL) shapes
  uint32_t added_cols[4]; // number of DDL-added columns per table
  int inited;
};

}  // extern "C"

namespace {

struct Gen {
  const synth_cfg& cfg; synth_state& st; Buf& out; std::vector<uint32_t>& offs;
  Buf msg{nullptr, 0};
  std::vector<uint8_t> scratch;

  void begin_msg() { scratch.resize(1 << 20); msg = Buf{scratch.data(), scratch.size()}; }
  // wraps msg as XLogData in a CopyData frame
  bool flush_w(uint64_t lsn) {
    size_t plen = 1 + 8 + 8 + 8 + msg.n;
    if (out.n + 5 + plen > out.cap) return false;
    out.u8('d'); out.be32((uint32_t)(plen + 4));
    out.u8('w'); out.be64(lsn); out.be64(lsn); out.be64(0);
    out.put(msg.p, msg.n);
    offs.push_back((uint32_t)out.n);
    st.frames_since_ka++;
    if (cfg.keepalive_every && st.frames_since_ka >= cfg.keepalive_every) {
      st.frames_since_ka = 0;
      if (out.n + 5 + 18 <= out.cap) {
        out.u8('d'); out.be32(18 + 4); out.u8('k'); out.be64(lsn); out.be64(0); out.u8(0);
        offs.push_back((uint32_t)out.n);
      }
    }
    return true;
  }
  uint64_t next_lsn() { st.lsn += 8; return st.lsn; }

  void text_cell(const char* s, size_t n) { msg.u8('t'); msg.be32((uint32_t)n); msg.put(s, n); }

  void gen_value(const synth_col& c, uint64_t row_id, bool allow_null = true) {
    Rng& r = st.rng;
    if (allow_null && c.nullable && c.null_pct && r.pct(c.null_pct)) { msg.u8('n'); return; }
    char b[8256];   // one value: texts up to 8 KiB (wide-row tiles of k_cells)
    int n = 0;
    switch (c.kind) {
      case CK_INT4_10D: n = snprintf(b, sizeof b, "%llu", (unsigned long long)r.range(1000000000ull, 2147483647ull)); break;
      case CK_INT8_SEQ: n = snprintf(b, sizeof b, "%llu", (unsigned long long)row_id); break;
      case CK_INT8: { int64_t v = (int64_t)r.next(); unsigned sh = (unsigned)r.below(40); n = snprintf(b, sizeof b, "%lld", (long long)(v >> sh)); break; }
      case CK_INT4: { int32_t v = (int32_t)(uint32_t)r.next(); unsigned sh = (unsigned)r.below(20); n = snprintf(b, sizeof b, "%d", v >> sh); break; }
      case CK_INT2: n = snprintf(b, sizeof b, "%d", (int)(int16_t)(uint16_t)r.next()); break;
      case CK_BOOL: b[0] = r.pct(50) ? 't' : 'f'; n = 1; break;
      case CK_NUMERIC: {
        unsigned nd = (unsigned)r.range(1, 12);
        if (r.pct(30)) b[n++] = '-';
        b[n++] = (char)('1' + r.below(9));
        for (unsigned i = 1; i < nd; i++) b[n++] = (char)('0' + r.below(10));
        b[n++] = '.'; b[n++] = (char)('0' + r.below(10)); b[n++] = (char)('0' + r.below(10));
        break;
      }
      case CK_TEXT: {
        unsigned len = (unsigned)r.range(c.min_len, c.max_len);
        if (len > 8192) len = 8192;
        while ((unsigned)n < len) {
          if (c.utf8_pct && (unsigned)n + 2 <= len && r.pct(c.utf8_pct)) {
            // a 2-byte UTF-8 char U+00A1..U+00FF / U+0100..U+017F
            unsigned cp = 0xA1 + (unsigned)r.below(0x17F - 0xA1);
            b[n++] = (char)(0xC0 | (cp >> 6)); b[n++] = (char)(0x80 | (cp & 63));
          } else {
            b[n++] = (char)(' ' + r.below(95));
          }
        }
        break;
      }
      case CK_TIMESTAMPTZ: {
        unsigned y = 2000 + (unsigned)r.below(40);
        unsigned mo = 1 + (unsigned)r.below(12);
        unsigned d = 1 + (unsigned)r.below(28);
        unsigned hh = (unsigned)r.below(24);
        unsigned mi = (unsigned)r.below(60);
        unsigned ss = (unsigned)r.below(60);
        unsigned us = (unsigned)r.below(1000000);
        n = snprintf(b, sizeof b, "%04u-%02u-%02u %02u:%02u:%02u.%06u+00", y, mo, d, hh, mi, ss, us);
        break;
      }
      case CK_UUID: {
        static const char* H = "0123456789abcdef";
        uint64_t a = r.next(), c2 = r.next();
        for (int i = 0; i < 36; i++) {
          if (i == 8 || i == 13 || i == 18 || i == 23) { b[n++] = '-'; continue; }
          uint64_t& w = n < 18 ? a : c2;
          b[n++] = H[w & 15]; w >>= 4;
        }
        break;
      }
      default: b[0] = '0'; n = 1; break;
    }
    text_cell(b, (size_t)n);
  }

  void tuple_full(const synth_table& t, uint64_t row_id, int toast_col = -1) {
    msg.be16((uint16_t)t.ncols);
    for (uint32_t i = 0; i < t.ncols; i++) {
      if ((int)i == toast_col) { msg.u8('u'); continue; }
      gen_value(t.cols[i], row_id);
    }
  }
  void tuple_key(const synth_table& t, uint64_t row_id, bool full_width) {
    uint32_t nk = 0;
    for (uint32_t i = 0; i < t.ncols; i++) nk += t.cols[i].pk;
    msg.be16((uint16_t)(full_width ? t.ncols : nk));
    for (uint32_t i = 0; i < t.ncols; i++) {
      if (t.cols[i].pk) gen_value(t.cols[i], row_id, false);
      else if (full_width) msg.u8('n');
    }
  }

  bool relation(const synth_table& t) {
    if (cfg.type_msg_pct && st.rng.pct(cfg.type_msg_pct)) {
      begin_msg(); msg.u8('Y'); msg.be32(90000 + t.rel_id % 100); msg.cstr("public"); msg.cstr("custom_type");
      if (!flush_w(next_lsn())) return false;
    }
    begin_msg();
    msg.u8('R'); msg.be32(t.rel_id); msg.cstr("public"); msg.cstr(t.name); msg.u8('d'); msg.be16((uint16_t)t.ncols);
    for (uint32_t i = 0; i < t.ncols; i++) {
      msg.u8(t.cols[i].pk ? 1 : 0); msg.cstr(t.cols[i].name); msg.be32(t.cols[i].type_oid); msg.be32(0xFFFFFFFFu);
    }
    return flush_w(next_lsn());
  }

  bool ddl(uint32_t ti) {
    synth_table& t = st.tables[ti];
    // alternate: add a column, then drop it again
    if (st.added_cols[ti] == 0 && t.ncols < 31) {
      synth_col c{};
      c.kind = (st.ddl_round & 1) ? CK_INT4 : CK_TEXT;
      c.type_oid = c.kind == CK_INT4 ? 23 : 25;
      c.nullable = 1; c.null_pct = 20; c.min_len = 1; c.max_len = 24;
      snprintf(c.name, sizeof c.name, "added_%u", st.ddl_round);
      t.cols[t.ncols++] = c;
      st.added_cols[ti] = 1;
    } else if (st.added_cols[ti]) {
      t.ncols--;
      st.added_cols[ti] = 0;
    }
    st.ddl_round++;
    std::string js = "{\"command_tag\":\"ALTER TABLE\",\"nspname\":\"public\",\"relname\":\"";
    js += t.name; js += "\",\"oid\":"; js += std::to_string(t.rel_id);
    js += ",\"identity\":{\"primary_key_attnums\":[";
    bool first = true;
    for (uint32_t i = 0; i < t.ncols; i++) if (t.cols[i].pk) { if (!first) js += ","; first = false; js += std::to_string(i + 1); }
    js += "],\"relreplident\":\"d\",\"replica_identity_index_attnums\":[]},\"extra_field_ignored\":true,\"columns\":[";
    for (uint32_t i = 0; i < t.ncols; i++) {
      const synth_col& c = t.cols[i];
      if (i) js += ",";
      js += "{\"attname\":\""; js += c.name; js += "\",\"atttypid\":"; js += std::to_string(c.type_oid);
      js += ",\"atttypmod\":-1,\"attnum\":"; js += std::to_string(i + 1);
      js += ",\"attnotnull\":"; js += c.nullable ? "false" : "true"; js += ",\"default_expression\":null}";
    }
    js += "]}";
    begin_msg();
    msg.u8('M'); msg.u8(1); msg.be64(st.lsn); msg.cstr("supabase_etl_ddl"); msg.be32((uint32_t)js.size()); msg.put(js.data(), js.size());
    if (!flush_w(next_lsn())) return false;
    return relation(t);
  }

  // One whole transaction; returns false (and rolls the output back) if it does not fit.
  bool txn(uint32_t rows) {
    size_t save_n = out.n, save_offs = offs.size();
    synth_state save = st;
    uint32_t ti = (uint32_t)(st.txn_no % cfg.ntables);
    uint64_t final_lsn = st.lsn + 8ull * (rows + 16) + 8;  // > every LSN inside the txn
    bool ok = true;
    begin_msg(); msg.u8('B'); msg.be64(final_lsn); msg.be64(700000000000000ll + (int64_t)st.txn_no); msg.be32((uint32_t)(1000 + st.txn_no));
    ok = flush_w(next_lsn());
    if (ok && cfg.emit_origin && !st.origin_sent) {
      begin_msg(); msg.u8('O'); msg.be64(st.lsn); msg.cstr("origin_a");
      ok = flush_w(next_lsn()); st.origin_sent = 1;
    }
    if (ok && cfg.ddl_every_txns && st.txn_no && st.txn_no % cfg.ddl_every_txns == 0) {
      ok = ddl(ti); st.rel_sent[ti] = 1;
    }
    const synth_table& t = st.tables[ti];
    if (ok && cfg.emit_relations && !st.rel_sent[ti]) { ok = relation(t); st.rel_sent[ti] = 1; }
    for (uint32_t r = 0; ok && r < rows; r++) {
      unsigned op = (unsigned)st.rng.below(100);
      uint64_t id = st.next_id++;
      begin_msg();
      if (op < cfg.pct_insert) {
        msg.u8('I'); msg.be32(t.rel_id); msg.u8('N'); tuple_full(t, id);
      } else if (op < cfg.pct_insert + cfg.pct_update) {
        msg.u8('U'); msg.be32(t.rel_id);
        unsigned shape = (unsigned)st.rng.below(100);
        int toast_col = -1;
        if (shape < cfg.upd_pct_key) { msg.u8('K'); tuple_key(t, id, true); }
        else if (shape < cfg.upd_pct_key + cfg.upd_pct_toast) {
          // SURVEY cfg3: carry 'u' on t2 — the second text column when there are
          // several, else the last non-key text column.
          int seen = 0;
          for (uint32_t i = 0; i < t.ncols; i++) {
            if (t.cols[i].kind != CK_TEXT || t.cols[i].pk) continue;
            toast_col = (int)i;
            if (++seen == 2) break;
          }
        }
        msg.u8('N'); tuple_full(t, id, toast_col);
      } else {
        msg.u8('D'); msg.be32(t.rel_id); msg.u8('K'); tuple_key(t, id, false);
      }
      ok = flush_w(next_lsn());
    }
    if (ok) {
      st.lsn = final_lsn - 8;
      begin_msg(); msg.u8('C'); msg.u8(0); msg.be64(final_lsn); msg.be64(final_lsn + 8); msg.be64(700000000000000ll + (int64_t)st.txn_no);
      ok = flush_w(next_lsn());
      st.lsn = final_lsn + 8;
    }
    if (!ok || out.overflow) { out.n = save_n; out.overflow = false; offs.resize(save_offs); st = save; return false; }
    st.txn_no++;
    return true;
  }
};

}  // namespace

extern "C" {

void synth_init(const synth_cfg* cfg, synth_state* st) {
  memset(st, 0, sizeof *st);
  st->rng.s = cfg->seed;
  st->lsn = cfg->start_lsn;
  st->next_id = 1;
  for (uint32_t i = 0; i < cfg->ntables && i < 4; i++) st->tables[i] = cfg->tables[i];
  st->inited = 1;
}

// Appends whole transactions to buf until `cap` bytes or `max_txns` are
// reached. offsets must hold max_frames+1 entries; offsets[0] = 0.
// Returns bytes written; *nframes = frames written.
size_t synth_fill(const synth_cfg* cfg, synth_state* st, uint8_t* buf, size_t cap, uint32_t* offsets,
                  size_t max_frames, uint64_t max_txns, size_t* nframes) {
  Buf out{buf, cap};
  std::vector<uint32_t> offs;
  offs.reserve(1024);
  offs.push_back(0);
  Gen g{*cfg, *st, out, offs, Buf{nullptr, 0}, {}};
  uint64_t done = 0;
  while (done < max_txns) {
    uint32_t rows = cfg->rows_per_txn;
    if (offs.size() + rows + 24 > max_frames) break;
    if (!g.txn(rows)) {
      // shrink the last transaction so the batch is filled close to cap
      bool fit = false;
      while (rows > 1) { rows /= 2; if (offs.size() + rows + 24 <= max_frames && g.txn(rows)) { fit = true; break; } }
      if (!fit) break;
    }
    done++;
  }
  memcpy(offsets, offs.data(), offs.size() * sizeof(uint32_t));
  *nframes = offs.size() - 1;
  return out.n;
}

size_t synth_cfg_size(void) { return sizeof(synth_cfg); }
size_t synth_state_size(void) { return sizeof(synth_state); }

}  // extern "C"
