// amwg_kernels.hip -- the step-kernel instantiations of ONE built-in model family (compiled once per family, -DAMWG_FAMILY=0..3, so the
// four families build in parallel): amwg_step_kernel<Model, G, BT> for every lane count G and every workgroup size class BT.
//
// BT is the register budget the instantiation is compiled for (__launch_bounds__): workgroups of up to 256 threads may use 512
// VGPRs per lane, 512 threads 256, 1024 threads 128.  A chain on G > 64 lanes is exactly one workgroup of G threads, so those have
// one class each.  The host (amwg_core.hip, choose_geometry) asks for the kernel of (G, workgroup size) through the lookup below.
#include <hip/hip_runtime.h>

#include "amwg_kernel.h"
#include "amwg_models.h"
#include "amwg_sampler.h"

using namespace amwg;

#if AMWG_FAMILY == 0
using Family = NormalModel;
#define AMWG_FAMILY_LOOKUP amwg_kernels_normal
#define AMWG_FAMILY_CERT_LOOKUP amwg_kernels_cert_normal
#elif AMWG_FAMILY == 1
using Family = BetaBernModel;
#define AMWG_FAMILY_LOOKUP amwg_kernels_beta_bern
#define AMWG_FAMILY_CERT_LOOKUP amwg_kernels_cert_beta_bern
#elif AMWG_FAMILY == 2
using Family = HierNormalModel;
#define AMWG_FAMILY_LOOKUP amwg_kernels_hier_normal
#define AMWG_FAMILY_CERT_LOOKUP amwg_kernels_cert_hier_normal
#elif AMWG_FAMILY == 3
using Family = PoisGlmModel;
#define AMWG_FAMILY_LOOKUP amwg_kernels_pois_glm
#define AMWG_FAMILY_CERT_LOOKUP amwg_kernels_cert_pois_glm
#else
#error "AMWG_FAMILY must be 0..3"
#endif

namespace {

#if defined(AMWG_X_BT1024)       // development experiment: every instantiation with the 1024-thread register budget, as in round 2
constexpr int class_of(int) { return 1024; }
#else
constexpr int class_of(int block) { return block <= 256 ? 256 : (block <= 512 ? 512 : 1024); }
#endif

template <int G>
step_kernel_t single_wave(int block) {      // G <= 64: any workgroup size up to the family's cap
  switch (class_of(block)) {
    case 256: return amwg_step_kernel<Family, G, 256>;
    case 512: if constexpr (Family::kMaxThreads >= 512) return amwg_step_kernel<Family, G, 512>; else return nullptr;
    default: if constexpr (Family::kMaxThreads >= 1024) return amwg_step_kernel<Family, G, 1024>; else return nullptr;
  }
}
template <int G>
step_kernel_t multi_wave(int block) {       // G > 64: one chain = one workgroup of G threads
  if (block != G) return nullptr;
  if constexpr (G <= Family::kMaxThreads) return amwg_step_kernel<Family, G, class_of(G)>;
  else return nullptr;
}

}  // namespace

#if AMWG_FAMILY == 2
// the group-local kernel of the hierarchical family (amwg_gl.h): a chain on one wavefront, any workgroup size class
step_kernel_t amwg_kernel_hier_gl(int block) {
  switch (class_of(block)) {
    case 256: return amwg_gl_kernel<HierGlModel, 256>;
    case 512: return amwg_gl_kernel<HierGlModel, 512>;
    default: return amwg_gl_kernel<HierGlModel, 1024>;
  }
}
#endif

#if AMWG_FAMILY == 2
// the sweep kernel of the hierarchical family (row layout, 64 lanes per chain: amwg_kernel.h kSweep)
step_kernel_t amwg_kernel_hier_sweep(int block) {
  switch (class_of(block)) {
    case 256: return amwg_sweep_kernel<HierNormalModel, 256>;
    case 512: return amwg_sweep_kernel<HierNormalModel, 512>;
    default: return amwg_sweep_kernel<HierNormalModel, 1024>;
  }
}
#endif

// the kernel that decides from certified values (amwg_kernel.h kCert; options.full_evaluation = 0), for the lane count the family has one at: the ordinary
// stepper (Normal: one lane per chain; Poisson: 16), or -- families whose certified value needs the row layout -- the sweep kernel (hierarchical: 64 lanes,
// workgroups of at most 512 threads).  nullptr: none.
namespace {
template <class Family>      // (a template: the branches a family has no kernel for must not be instantiated)
step_kernel_t certified_lookup(int lanes, int block) {
  if constexpr (CertifiedOf<Family>::value) {
    constexpr int GC = CertifiedOf<Family>::lanes;
    if (lanes != GC) return nullptr;
    if constexpr (CertNeedsRows<Family>::value) {
      switch (class_of(block)) {
        case 256: return amwg_sweep_kernel_cert<Family, 256>;
        case 512: return amwg_sweep_kernel_cert<Family, 512>;
        default: return nullptr;
      }
    } else {
      switch (class_of(block)) {
        case 256: return amwg_step_kernel_cert<Family, GC, 256>;
        case 512: if constexpr (Family::kMaxThreads >= 512) return amwg_step_kernel_cert<Family, GC, 512>; else return nullptr;
        default: if constexpr (Family::kMaxThreads >= 1024) return amwg_step_kernel_cert<Family, GC, 1024>; else return nullptr;
      }
    }
  }
  (void)lanes; (void)block;
  return nullptr;
}
}  // namespace
step_kernel_t AMWG_FAMILY_CERT_LOOKUP(int lanes, int block) { return certified_lookup<Family>(lanes, block); }

step_kernel_t AMWG_FAMILY_LOOKUP(int lanes, int block) {
  switch (lanes) {
    case 1: return single_wave<1>(block);
    case 2: return single_wave<2>(block);
    case 4: return single_wave<4>(block);
    case 8: return single_wave<8>(block);
    case 16: return single_wave<16>(block);
    case 32: return single_wave<32>(block);
    case 64: return single_wave<64>(block);
    case 128: return multi_wave<128>(block);
    case 256: return multi_wave<256>(block);
    case 512: return multi_wave<512>(block);
    case 1024: return multi_wave<1024>(block);
  }
  return nullptr;
}
