'use strict';
// translate_cli.js -- test helper: translates one closure of tests/js/user_models.js with the PRODUCT's
// translator and writes <out>/<name>.hip (source), <name>.arrays.bin (u32 count, then per array u64 len + f64 data)
// and <name>.meta.json.   node tests/js/translate_cli.js <outdir> [name ...]
const fs = require('fs');
const path = require('path');
const { mcmc, ld } = require('../../bayes.js_amd');
const um = require('./user_models.js');
global.ld = ld;
const out = process.argv[2];
const want = process.argv.slice(3);
for (const name of (want.length ? want : um.names)) {
  const m = um.build(name);
  const params = mcmc.complete_params(m.params, mcmc.param_init_fixed);
  // ($AMWG_TRANSLATE_OPTS: extra translator options as JSON -- development A/B runs, e.g. {"no_open_softplus":true,"max_threads":256})
  const tr = mcmc.translate(m.log_post, params, m.data, Object.assign({ helpers: m.helpers, constants: m.constants }, JSON.parse(process.env.AMWG_TRANSLATE_OPTS || '{}')));
  fs.writeFileSync(path.join(out, name + '.hip'), tr.source);
  let bytes = 4;
  for (const a of tr.arrays) bytes += 8 + a.length * 8;
  const buf = Buffer.alloc(bytes);
  let o = 0;
  buf.writeUInt32LE(tr.arrays.length, o); o += 4;
  for (const a of tr.arrays) {
    buf.writeBigUInt64LE(BigInt(a.length), o); o += 8;
    for (let i = 0; i < a.length; i++) { buf.writeDoubleLE(a[i], o); o += 8; }
  }
  fs.writeFileSync(path.join(out, name + '.arrays.bin'), buf);
  fs.writeFileSync(path.join(out, name + '.meta.json'), JSON.stringify({ name, P: tr.P, derived: tr.derived, lds_bytes: tr.lds_bytes, lds_bytes_one_lane: tr.lds_bytes_one_lane,
    parallel: tr.parallel, max_threads: tr.max_threads, work_per_eval: tr.work_per_eval, work_one_lane: tr.work_one_lane, rows_n_obs: tr.rows_n_obs, rows_groups: tr.rows_groups, rows_sweep: tr.rows_sweep, cert_tail_n: tr.cert_tail_n, rows_cert: tr.rows_cert, pois_tail_n: tr.pois_tail_n, array_keys: tr.array_keys, array_types: tr.array_types, array_len: tr.arrays.map((a) => a.length) }));
}
