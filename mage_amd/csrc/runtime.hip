// Library state: thread-local error string, the per-device zero page used for padded gather loads, and the per-device
// deferred-error word that kernels raise when they meet an argument only visible on the device (an out-of-range id).
#include "common.h"
#include <string.h>

static thread_local char g_err[512] = "";
static void* g_zero[MAGE_MAX_DEVICES] = {nullptr};
static int* g_flag[MAGE_MAX_DEVICES] = {nullptr};      // [0] code (0 = none), [1] low 32 bits of the offending value, [2] table size

void mage_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

int mage_device_index() {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= MAGE_MAX_DEVICES) return -1;
    return dev;
}

const void* mage_zero_page() {
    const int dev = mage_device_index();
    return dev < 0 ? nullptr : g_zero[dev];
}

int* mage_error_word() {
    const int dev = mage_device_index();
    return dev < 0 ? nullptr : g_flag[dev];
}

extern "C" int mage_abi_version(void) { return MAGE_ABI_VERSION; }

extern "C" const char* mage_last_error(void) { return g_err; }

extern "C" int mage_init(int device) {
    if (device < 0 || device >= MAGE_MAX_DEVICES) {
        mage_set_error("mage_init: device %d out of range", device);
        return MAGE_EINVAL;
    }
    if (g_zero[device]) return MAGE_OK;
    // architecture first: nothing is allocated (and nothing is remembered) on a device this library cannot run on
    hipDeviceProp_t p;
    hipError_t e = hipGetDeviceProperties(&p, device);
    if (e == hipSuccess && strncmp(p.gcnArchName, "gfx950", 6) != 0) {
        mage_set_error("mage_init: device %d is %s; this library is built for gfx950 (MI355X) only", device, p.gcnArchName);
        return MAGE_EUNSUPPORTED;
    }
    int prev = -1;
    void* zero = nullptr;
    int* flag = nullptr;
    if (e == hipSuccess) e = hipGetDevice(&prev);
    if (e == hipSuccess) e = hipSetDevice(device);
    if (e == hipSuccess) e = hipMalloc(&zero, 16384);
    if (e == hipSuccess) e = hipMemset(zero, 0, 16384);
    if (e == hipSuccess) e = hipMalloc((void**)&flag, 16);
    if (e == hipSuccess) e = hipMemset(flag, 0, 16);
    if (e == hipSuccess) e = hipDeviceSynchronize();
    if (prev >= 0 && prev != device) (void)hipSetDevice(prev);        // the caller's current device is left as it was
    if (e != hipSuccess) {
        if (zero) (void)hipFree(zero);
        if (flag) (void)hipFree(flag);
        mage_set_error("mage_init: %s", hipGetErrorString(e));
        return MAGE_EHIP;
    }
    g_flag[device] = flag;
    g_zero[device] = zero;
    return MAGE_OK;
}

extern "C" int mage_check_device_errors(void* stream) {
    int* flag = mage_error_word();
    MAGE_CHECK_ARG(flag != nullptr, "mage_check_device_errors: mage_init() has not been called on the current device");
    int host[4] = {0, 0, 0, 0};
    hipError_t e = hipMemcpyAsync(host, flag, 16, hipMemcpyDeviceToHost, (hipStream_t)stream);
    if (e == hipSuccess) e = hipStreamSynchronize((hipStream_t)stream);
    if (e != hipSuccess) {
        mage_set_error("mage_check_device_errors: %s", hipGetErrorString(e));
        return MAGE_EHIP;
    }
    if (host[0] == 0) return MAGE_OK;
    (void)hipMemsetAsync(flag, 0, 16, (hipStream_t)stream);
    const char* what = host[0] == MAGE_DEVERR_EMBEDDING_ID ? "mage_embedding: index out of range"
                       : host[0] == MAGE_DEVERR_CE_TARGET ? "mage_cross_entropy: target out of range"
                                                          : "device-side argument error";
    mage_set_error("%s (value %d, valid range [0, %d))", what, host[1], host[2]);
    return MAGE_EINVAL;
}
