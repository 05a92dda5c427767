// Compute-unit budget for data-parallel overlap: a CU-masked compute stream and the grid size of the persistent kernels
// (include/libra_hip.h, "compute-unit budget").
#include <atomic>
#include <vector>
#include "hip_common.hpp"
#include "../../include/libra_hip.h"

namespace libra {

static std::atomic<int> g_cu_budget{0};
static std::atomic<int> g_cu_count{0};

int cu_count() {
    int n = g_cu_count.load();
    if (!n) {
        int dev = 0, cus = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) cus = 256;
        g_cu_count = n = cus;                     // (benign race: every thread stores the same value)
    }
    return n;
}

// persistent grid: at most one workgroup per budgeted CU, rounded down to a multiple of the rotation period (the kernels' static
// schedules need it), at least one period, at most one workgroup per item
long persistent_grid(long nitems, int period) {
    int cus = cu_count();
    const int b = g_cu_budget.load();
    if (b > 0 && b < cus) cus = b;
    long nblk = (long)cus / period * period;
    if (nblk < period) nblk = period;
    return nblk > nitems ? nitems : nblk;
}

}  // namespace libra

extern "C" int libra_get_cu_count(void) { return libra::cu_count(); }

extern "C" int libra_set_cu_budget(int32_t cus) {
    if (cus < 0) return LIBRA_ERR_SHAPE;
    return libra::g_cu_budget.exchange(cus);
}

extern "C" int libra_stream_create_cu_reserved(int32_t reserve_cus, void** stream_out, int32_t* cus_out) {
    if (!stream_out) return LIBRA_ERR_ALIGN;
    const int cus = libra::cu_count();
    if (reserve_cus < 0 || reserve_cus >= cus) return LIBRA_ERR_SHAPE;
    const int use = cus - reserve_cus;
    // Mask layout on MI355X (measured, profiles/r05_cu_mask_layout.txt): bit b is a CU of XCC b % 8; inside the XCC, index b / 8 is
    // shader engine (b / 8) % 4, CU slot (b / 8) / 4.  Clearing the highest `reserve_cus` bits therefore takes the reserved CUs one per
    // XCC in turn, and inside an XCC from SE 3 downwards - SE-symmetric only at multiples of 32.  (A word of all zeros would leave its
    // XCC unrestricted; reserve_cus < cus keeps every XCC's bits non-empty for the reserves that make sense.)
    std::vector<uint32_t> mask((cus + 31) / 32, 0u);
    for (int i = 0; i < use; ++i) mask[i >> 5] |= 1u << (i & 31);
    hipStream_t s = nullptr;
    if (hipExtStreamCreateWithCUMask(&s, (uint32_t)mask.size(), mask.data()) != hipSuccess) return LIBRA_ERR_LAUNCH;
    *stream_out = (void*)s;
    if (cus_out) *cus_out = use;
    return LIBRA_OK;
}

extern "C" int libra_stream_destroy(void* stream) {
    if (!stream) return LIBRA_OK;
    return hipStreamDestroy((hipStream_t)stream) == hipSuccess ? LIBRA_OK : LIBRA_ERR_LAUNCH;
}

// ---- diagnostics: which physical CUs a CU mask enables (tools/cu_mask_probe.py; the layout of the mask is not documented for gfx950)
namespace libra {
__global__ __launch_bounds__(64) void cu_map_kernel(unsigned* out, int spin) {
    extern __shared__ char hold[];                       // 128 KiB requested at launch: one workgroup per CU
    unsigned hw, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    for (int i = 0; i < spin; ++i) __builtin_amdgcn_s_sleep(64);      // stay resident long enough for every enabled CU to receive a workgroup
    if (threadIdx.x == 0) { out[2 * blockIdx.x] = hw; out[2 * blockIdx.x + 1] = xcc; hold[0] = 0; }
}
}  // namespace libra

extern "C" int libra_stream_create_cu_mask(const uint32_t* mask, int32_t nwords, void** stream_out) {
    if (!mask || !stream_out || nwords <= 0) return LIBRA_ERR_ALIGN;
    hipStream_t s = nullptr;
    if (hipExtStreamCreateWithCUMask(&s, (uint32_t)nwords, mask) != hipSuccess) return LIBRA_ERR_LAUNCH;
    *stream_out = (void*)s;
    return LIBRA_OK;
}

extern "C" int libra_debug_cu_map(uint32_t* out, int32_t nblocks, int32_t spin, void* stream) {
    if (!out || nblocks <= 0) return LIBRA_ERR_ALIGN;
    (void)hipFuncSetAttribute((const void*)libra::cu_map_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
    hipLaunchKernelGGL(libra::cu_map_kernel, dim3((unsigned)nblocks), dim3(64), 128 * 1024, (hipStream_t)stream, out, (int)spin);
    return hipGetLastError() == hipSuccess ? LIBRA_OK : LIBRA_ERR_LAUNCH;
}
