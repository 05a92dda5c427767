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
