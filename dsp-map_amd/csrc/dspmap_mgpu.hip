// dspmap_mgpu.hip -- split-phase frame for Z-slab sharding across GPUs (one
// process per GPU; collectives are issued by the caller through
// torch.distributed / RCCL).  Round-1 status: the slab-aware kernels exist
// (every kernel takes z_lo/z_hi through MapDims and k_predict marks particles
// that leave the slab), the exchange entry points below are not wired yet and
// report an error instead of silently doing nothing.
#include <hip/hip_runtime.h>
#include "../../include/dspmap.h"

static int not_yet(const char* name) {
    fprintf(stderr, "libdspmap_hip: %s is not implemented yet\n", name);
    return DSPMAP_E_STATE;
}
extern "C" int dspmap_mgpu_begin(dspmap_t*, int, const float*, int, const dspmap_vpoint*, const float*, double, const float*) { return not_yet("dspmap_mgpu_begin"); }
extern "C" int dspmap_mgpu_get_exports(dspmap_t*, int, const float**, int*) { return not_yet("dspmap_mgpu_get_exports"); }
extern "C" int dspmap_mgpu_import_movers(dspmap_t*, int, const float*) { return not_yet("dspmap_mgpu_import_movers"); }
extern "C" int dspmap_mgpu_ck_partial(dspmap_t*, float**, int*) { return not_yet("dspmap_mgpu_ck_partial"); }
extern "C" int dspmap_mgpu_nstatic_partial(dspmap_t*, int**, int*) { return not_yet("dspmap_mgpu_nstatic_partial"); }
extern "C" int dspmap_mgpu_finish(dspmap_t*) { return not_yet("dspmap_mgpu_finish"); }
