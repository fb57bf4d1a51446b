// Library state: thread-local error string and the per-device zero page used for padded gather loads.
#include "common.h"
#include <string.h>

static thread_local char g_err[512] = "";
static void* g_zero[16] = {nullptr};

void mage_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

const void* mage_zero_page() {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return nullptr;
    return g_zero[dev];
}

extern "C" int mage_abi_version(void) { return MAGE_ABI_VERSION; }

extern "C" const char* mage_last_error(void) { return g_err; }

extern "C" int mage_init(int device) {
    if (device < 0 || device >= 16) {
        mage_set_error("mage_init: device %d out of range", device);
        return MAGE_EINVAL;
    }
    if (g_zero[device]) return MAGE_OK;
    hipError_t e = hipSetDevice(device);
    if (e == hipSuccess) e = hipMalloc(&g_zero[device], 4096);
    if (e == hipSuccess) e = hipMemset(g_zero[device], 0, 4096);
    if (e == hipSuccess) {
        hipDeviceProp_t p;
        e = hipGetDeviceProperties(&p, device);
        if (e == hipSuccess && strncmp(p.gcnArchName, "gfx950", 6) != 0) {
            mage_set_error("mage_init: device %d is %s; this library is built for gfx950 (MI355X) only", device, p.gcnArchName);
            return MAGE_EUNSUPPORTED;
        }
    }
    if (e != hipSuccess) {
        g_zero[device] = nullptr;
        mage_set_error("mage_init: %s", hipGetErrorString(e));
        return MAGE_EHIP;
    }
    return MAGE_OK;
}
