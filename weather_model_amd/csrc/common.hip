// common.hip -- error text, version and device queries of libwm.so.
#include "common.h"

#include <mutex>

namespace {
thread_local char g_err[512] = "";
std::mutex g_dev_mutex;
int g_cus[64] = {0};
}  // namespace

void wm_set_error(const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

int wm_num_cus()
{
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 0;
    std::lock_guard<std::mutex> lock(g_dev_mutex);
    if (g_cus[dev] == 0) {
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, dev) != hipSuccess) return 0;
        g_cus[dev] = prop.multiProcessorCount;
    }
    return g_cus[dev];
}

extern "C" int wm_version(void) { return WM_VERSION_NUM; }

extern "C" const char *wm_last_error(void) { return g_err; }

extern "C" int wm_device_cus(void)
{
    const int c = wm_num_cus();
    if (c <= 0) {
        wm_set_error("wm_device_cus: no HIP device available");
        return WM_ERR_NO_DEVICE;
    }
    return c;
}
