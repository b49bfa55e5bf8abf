// Translation unit of encdec_persistent_kernel (namp_kernels.h): the whole encoder + decoder pass of a small batch as one
// launch.  Separate from namp.hip because its six inlined stages take minutes to compile; namp.hip builds the PersistArgs
// and calls namp_internal_launch_persistent.
#include "namp_kernels.h"

#include <mutex>

namespace {
std::once_flag g_once;
hipError_t g_err = hipSuccess;
void set_attrs() {
  auto set = [](const void* f) {
    hipError_t e = hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, EDGE_TAIL_LDS);
    if (e != hipSuccess) g_err = e;
  };
  set((const void*)(encdec_persistent_kernel<4, PREC_F32>));
  set((const void*)(encdec_persistent_kernel<4, PREC_X3>));
}
}  // namespace

static_assert(sizeof(PersistArgs) <= 4000, "PersistArgs must fit the 4 KiB kernel-argument segment");

__attribute__((visibility("hidden"))) int namp_internal_launch_persistent(const PersistArgs* a, int x3, int grid, int block, hipStream_t s) {
  std::call_once(g_once, set_attrs);
  if (g_err != hipSuccess) return (int)g_err;
  if (x3) hipLaunchKernelGGL((encdec_persistent_kernel<4, PREC_X3>), dim3(grid), dim3(block), EDGE_TAIL_LDS, s, *a);
  else hipLaunchKernelGGL((encdec_persistent_kernel<4, PREC_F32>), dim3(grid), dim3(block), EDGE_TAIL_LDS, s, *a);
  return 0;
}
