'use strict';
/*
 * philox.js -- TEST INFRASTRUCTURE ONLY.
 * JavaScript twin of the Philox4x32-10 counter-based generator (Salmon et al.,
 * "Parallel random numbers: as easy as 1, 2, 3", SC'11) that the HIP kernel
 * uses per chain.  It exists so the *unmodified* reference sampler can be run
 * with a seeded, reproducible stream: the reference draws all randomness from
 * Math.random() looked up at call time (mcmc.js:32,37,46,47,230,528,762), so
 * assigning Math.random = stream(seed, chain) seeds it.
 *
 * Stream contract (shared with oracle/amwg_oracle.c and csrc/amwg_philox.h):
 *   block b of chain c under seed s = philox4x32_10(ctr = {b_lo, b_hi, c_lo, c_hi},
 *                                                   key = {s_lo, s_hi}) -> r0..r3
 *   uniform #2b   = ((r0 * 2^21) + (r1 >>> 11)) * 2^-53
 *   uniform #2b+1 = ((r2 * 2^21) + (r3 >>> 11)) * 2^-53         (both in [0,1))
 */
const M0 = 0xD2511F53, M1 = 0xCD9E8D57, W0 = 0x9E3779B9, W1 = 0xBB67AE85;

// 32x32 -> high 32 bits, via 16-bit limbs (no BigInt in the hot loop)
function mulhi(a, b) {
  const a0 = a & 0xffff, a1 = a >>> 16, b0 = b & 0xffff, b1 = b >>> 16;
  const p00 = a0 * b0, p01 = a0 * b1, p10 = a1 * b0, p11 = a1 * b1;
  const mid = (p00 >>> 16) + (p01 & 0xffff) + (p10 & 0xffff);
  return (p11 + (p01 >>> 16) + (p10 >>> 16) + (mid >>> 16)) >>> 0;
}

function philox4x32_10(c0, c1, c2, c3, k0, k1) {
  for (let r = 0; r < 10; r++) {
    const hi0 = mulhi(M0, c0), lo0 = Math.imul(M0, c0) >>> 0;
    const hi1 = mulhi(M1, c2), lo1 = Math.imul(M1, c2) >>> 0;
    const n0 = (hi1 ^ c1 ^ k0) >>> 0, n2 = (hi0 ^ c3 ^ k1) >>> 0;
    c0 = n0; c1 = lo1; c2 = n2; c3 = lo0;
    k0 = (k0 + W0) >>> 0; k1 = (k1 + W1) >>> 0;
  }
  return [c0, c1, c2, c3];
}

const TWO_M53 = 1.1102230246251565e-16; // 2^-53

function splitU64(x) { // accepts Number (<2^53) or BigInt
  const b = BigInt(x);
  return [Number(b & 0xffffffffn), Number((b >> 32n) & 0xffffffffn)];
}

/** Returns a function() -> uniform in [0,1); .count = uniforms consumed so far,
 *  .last = last value returned. `start` = index of the first uniform to return. */
function stream(seed, chain, start) {
  const [k0, k1] = splitU64(seed), [c2, c3] = splitU64(chain);
  let n = start || 0;      // uniforms consumed (exact below 2^53)
  let blk = -1, w = null;
  const f = function () {
    const b = Math.floor(n / 2);
    if (b !== blk) {
      w = philox4x32_10(b >>> 0, Math.floor(b / 4294967296) >>> 0, c2, c3, k0, k1);
      blk = b;
    }
    const h = (n % 2) * 2;
    n++;
    f.count = n;
    f.last = (w[h] * 2097152 + (w[h + 1] >>> 11)) * TWO_M53;
    return f.last;
  };
  f.count = n;
  f.last = NaN;
  return f;
}

module.exports = { philox4x32_10, stream, mulhi };
