// libhubbardtn_hip.so -- hand-written CDNA4 (gfx950) kernels behind the C ABI of
// include/hubbardtn_hip.h.  No CPU path exists in this library: every compute entry point
// launches HIP kernels.  See DESIGN.md for the data layout and the roofline bounding each kernel.
#include "htn_common.h"

extern "C" int htn_device_init(int device, char* name_host, int* cu_count_host) {
    HIP_TRY(hipSetDevice(device));
    hipDeviceProp_t p;
    HIP_TRY(hipGetDeviceProperties(&p, device));
    if (name_host) {
        strncpy(name_host, p.gcnArchName, 255);
        name_host[255] = 0;
    }
    if (cu_count_host) *cu_count_host = p.multiProcessorCount;
    if (strncmp(p.gcnArchName, "gfx950", 6) != 0)
        return fail_msg("htn_device_init: this library is built for gfx950 (MI355X) only");
    return 0;
}
