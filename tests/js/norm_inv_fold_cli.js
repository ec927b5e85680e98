const t = require(require('path').join(__dirname, '..', '..', 'bayes.js_amd', 'translate.js'));
const sds = [10, 100, 1, 0.5, 3, 2.5, 1e-3, 1e3, 7.25, 0.1, 1e10, 1e-10, 123456.789, Math.PI, 1/3];
let s = 77; const r = () => { s = (s * 1103515245 + 12345) % 2147483648; return s / 2147483648; };
for (let i = 0; i < 300; i++) sds.push(Math.exp((r() - 0.5) * 60));
for (const sd of sds) console.log(t.hexFloat(sd) + ' ' + t.foldConstantNormInv('norm_inv(' + t.hexFloat(sd) + ')'));
