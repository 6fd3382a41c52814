// Version and status strings of libpcb_hip.so (see include/pcb_hip.h).
#include "pcb_common.h"

extern "C" int pcb_version(void) { return 100; }

extern "C" const char *pcb_status_string(int status)
{
    switch (status) {
        case PCB_OK: return "ok";
        case PCB_ERR_INVALID_ARG: return "invalid argument";
        case PCB_ERR_UNSUPPORTED: return "size not supported by this build";
        case PCB_ERR_LAUNCH: return "kernel launch failed";
        default: return "unknown status";
    }
}
