'use strict';
/*
 * ref_dir.js -- TEST INFRASTRUCTURE ONLY: where the UNMODIFIED reference (mcmc.js + distributions.js) can be loaded from.
 *   1. $AMWG_REF_DIR                     (explicit)
 *   2. /root/reference                   (the build container)
 *   3. oracle/_ref                       (made from 2. by `make -C oracle ref`; git-ignored, travels to the GPU box with the snapshot)
 * Returns null when none has mcmc.js.  sha256() gives the hashes of the two files so that callers can check them against the
 * pinned oracle/ref.sha256 (the copy on the GPU box must be the reference, not something edited).
 */
const fs = require('fs'), path = require('path'), crypto = require('crypto');
function candidates() {
  const c = [];
  if (process.env.AMWG_REF_DIR) c.push(process.env.AMWG_REF_DIR);
  c.push('/root/reference', path.join(__dirname, '_ref'));
  return c;
}
function refDir() {
  for (const d of candidates()) if (fs.existsSync(path.join(d, 'mcmc.js')) && fs.existsSync(path.join(d, 'distributions.js'))) return d;
  return null;
}
function sha256(dir) {
  const out = {};
  for (const f of ['mcmc.js', 'distributions.js']) out[f] = crypto.createHash('sha256').update(fs.readFileSync(path.join(dir, f))).digest('hex');
  return out;
}
function pinned() {
  const out = {};
  for (const line of fs.readFileSync(path.join(__dirname, 'ref.sha256'), 'utf8').split('\n')) {
    const m = /^([0-9a-f]{64})\s+(\S+)$/.exec(line.trim());
    if (m) out[m[2]] = m[1];
  }
  return out;
}
function unmodified(dir) {
  const have = sha256(dir), want = pinned();
  return Object.keys(want).every((f) => have[f] === want[f]);
}
module.exports = { refDir, sha256, pinned, unmodified };
