// user_eval_host.cpp -- TEST INFRASTRUCTURE: evaluates a translated closure (the generated
// `amwg::UserModel`, the same text hiprtc compiles for the GPU) on the host, in the summation order
// of G lanes per chain (lane partial sums + xor butterfly, as csrc/amwg_kernel.h does).
//   g++ -std=c++17 -O2 -ffp-contract=off -I bayes.js_amd/csrc -DAMWG_USER_SOURCE='"<file.hip>"' -shared -fPIC ...
#include "amwg_user.h"
#include AMWG_USER_SOURCE

namespace {
template <int G>
double lanes(const double *state, const amwg::DataRef &d, double *dv) {
  const amwg::StateView S{state};
  double acc[G];
  for (int sub = 0; sub < G; ++sub) acc[sub] = amwg::UserModel::eval<G, false>(S, d, nullptr, sub, nullptr);
  for (int off = 1; off < G; off <<= 1) {
    double t[G];
    for (int j = 0; j < G; ++j) t[j] = acc[j] + acc[j ^ off];
    for (int j = 0; j < G; ++j) acc[j] = t[j];
  }
  if (dv) (void)amwg::UserModel::eval<G, true>(S, d, nullptr, 0, dv);
  return acc[0];
}
}  // namespace

extern "C" int user_num_derived() { return amwg::UserModel::kDerived; }

// arrays[j]: device-typed storage (f64 / u8 / i32 as the translator chose, see meta.array_types)
extern "C" double user_eval(const double *state, const void *const *arrays, int n_arrays, int G, double *dv) {
  amwg::DataRef d{};
  for (int j = 0; j < n_arrays && j < amwg::kInlineUserArrays; ++j) d.arr[j] = arrays[j];
  d.arr_ext = n_arrays > amwg::kInlineUserArrays ? arrays + amwg::kInlineUserArrays : nullptr;   // same split as the device (user_arr)
  switch (G) {
    case 1: return lanes<1>(state, d, dv);
    case 2: return lanes<2>(state, d, dv);
    case 4: return lanes<4>(state, d, dv);
    case 8: return lanes<8>(state, d, dv);
    case 16: return lanes<16>(state, d, dv);
    case 32: return lanes<32>(state, d, dv);
    case 64: return lanes<64>(state, d, dv);
    case 128: return lanes<128>(state, d, dv);
    case 256: return lanes<256>(state, d, dv);
    case 512: return lanes<512>(state, d, dv);
    case 1024: return lanes<1024>(state, d, dv);
  }
  return __builtin_nan("");
}
