// Library state: thread-local error string, the per-device zero page used for padded gather loads, and the per-device
// deferred-error word that kernels raise when they meet an argument only visible on the device (an out-of-range id).
#include <mutex>
#include "common.h"
#include <string.h>
#include <stdlib.h>

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

// ---- options: one table, read once from the environment, changed through mage_set_option
static MageOptions g_opt;
static std::once_flag g_opt_once;             // the first mage_gemm calls of several host threads race for the table otherwise
struct OptField { const char* name; int MageOptions::*field; };
static const OptField g_opt_fields[] = {
    {"gemm_no_4w", &MageOptions::gemm_no_4w}, {"gemm_no_4h", &MageOptions::gemm_no_4h}, {"gemm_4h_plain", &MageOptions::gemm_4h_plain}, {"gemm4_train_forms", &MageOptions::gemm4_train_forms},
    {"gemm_no_8phase", &MageOptions::gemm_no_8phase}, {"gemm_no_taps8", &MageOptions::gemm_no_taps8},
    {"gemm_no_narrow", &MageOptions::gemm_no_narrow}, {"gemm_no_narrow_few", &MageOptions::gemm_no_narrow_few},
    {"gemm_no_small", &MageOptions::gemm_no_small}, {"gemm_small_m", &MageOptions::gemm_small_m},
    {"gemm_stagger_groups", &MageOptions::gemm_stagger_groups},
    {"gemm_stagger_percent", &MageOptions::gemm_stagger_percent}, {"gemm_stagger_forced", &MageOptions::gemm_stagger_forced},
    {"gemm4_stagger_groups", &MageOptions::gemm4_stagger_groups}, {"gemm4_stagger_percent", &MageOptions::gemm4_stagger_percent},
    {"attn_no_mfma", &MageOptions::attn_no_mfma}, {"attn_no_fewq", &MageOptions::attn_no_fewq}, {"vq_no_mfma", &MageOptions::vq_no_mfma}, {"conv_no_tile", &MageOptions::conv_no_tile},
};
static int env_flag(const char* name) {
    const char* e = getenv(name);
    return (e && *e && strcmp(e, "0") != 0) ? 1 : 0;
}
const MageOptions& mage_options() {
    std::call_once(g_opt_once, [] {
        MageOptions o = {};
        o.gemm_no_4w = env_flag("MAGE_GEMM_NO_4W");
        o.gemm_no_4h = env_flag("MAGE_GEMM_NO_4H");
        o.gemm_4h_plain = env_flag("MAGE_GEMM_4H_PLAIN");
        o.gemm4_train_forms = env_flag("MAGE_GEMM4_TRAIN_FORMS");
        o.gemm_no_8phase = env_flag("MAGE_GEMM_NO_8PHASE");
        o.gemm_no_taps8 = env_flag("MAGE_GEMM_NO_TAPS8");
        o.gemm_no_narrow = env_flag("MAGE_GEMM_NO_NARROW");
        o.gemm_no_narrow_few = env_flag("MAGE_GEMM_NO_NARROW_FEW");
        o.gemm_no_small = env_flag("MAGE_GEMM_NO_SMALL");
        o.gemm_small_m = getenv("MAGE_GEMM_SMALL_M") ? atoi(getenv("MAGE_GEMM_SMALL_M")) : 1024;
        o.gemm_stagger_groups = 8;
        o.gemm_stagger_percent = 60;
        if (const char* e = getenv("MAGE_GEMM_STAGGER")) {
            o.gemm_stagger_forced = 1;
            if (sscanf(e, "%d,%d", &o.gemm_stagger_groups, &o.gemm_stagger_percent) < 2) o.gemm_stagger_percent = 60;
            if (o.gemm_stagger_groups < 0) o.gemm_stagger_groups = 0;
        }
        o.gemm4_stagger_groups = 8;
        o.gemm4_stagger_percent = 100;
        if (const char* e = getenv("MAGE_GEMM4_STAGGER")) {
            if (sscanf(e, "%d,%d", &o.gemm4_stagger_groups, &o.gemm4_stagger_percent) < 2) o.gemm4_stagger_percent = 100;
            if (o.gemm4_stagger_groups < 0) o.gemm4_stagger_groups = 0;
        }
        o.attn_no_mfma = env_flag("MAGE_ATTN_NO_MFMA");
        o.attn_no_fewq = env_flag("MAGE_ATTN_NO_FEWQ");
        o.vq_no_mfma = env_flag("MAGE_VQ_NO_MFMA");
        o.conv_no_tile = env_flag("MAGE_CONV_NO_TILE");
        if (o.gemm_small_m < 0) o.gemm_small_m = 0;
        g_opt = o;
    });
    return g_opt;
}
extern "C" int mage_set_option(const char* name, int32_t value) {
    MAGE_CHECK_ARG(name != nullptr, "mage_set_option: null name");
    (void)mage_options();
    // values: switches are 0 / 1, counts and percentages have ranges (a negative row bound or group count would index nothing sensible).
    // Changing an option while another host thread is inside a dispatch function is the caller's race: set options before the threads start.
    const bool is_count = strstr(name, "stagger_groups") != nullptr, is_percent = strstr(name, "stagger_percent") != nullptr;
    const bool is_rows = strcmp(name, "gemm_small_m") == 0;
    for (const OptField& f : g_opt_fields)
        if (strcmp(f.name, name) == 0) {
            MAGE_CHECK_ARG(is_count ? (value >= 0 && value <= 64) : is_percent ? (value >= 0 && value <= 400) : is_rows ? (value >= 0 && value <= (1 << 20))
                                                                                                                    : (value == 0 || value == 1),
                           "mage_set_option: value %d out of range for '%s' (switches: 0 | 1; stagger groups 0..64, percent 0..400; gemm_small_m 0..2^20)",
                           (int)value, name);
            g_opt.*(f.field) = value;
            return MAGE_OK;
        }
    mage_set_error("mage_set_option: unknown option '%s'", name);
    return MAGE_EINVAL;
}
extern "C" int mage_get_option(const char* name, int32_t* value) {
    MAGE_CHECK_ARG(name != nullptr && value != nullptr, "mage_get_option: null argument");
    const MageOptions& o = mage_options();
    for (const OptField& f : g_opt_fields)
        if (strcmp(f.name, name) == 0) {
            *value = o.*(f.field);
            return MAGE_OK;
        }
    mage_set_error("mage_get_option: unknown option '%s'", name);
    return MAGE_EINVAL;
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
