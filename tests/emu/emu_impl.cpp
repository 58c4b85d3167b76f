// TEST INFRASTRUCTURE: storage for the emulated built-in variables + stub for the (inline-asm) microbenchmarks.
#define GA_HIP_EMULATION_IMPL
#include <hip/hip_runtime.h>
#include <stdio.h>
namespace ga {
struct Ctx;
int util_microbench(Ctx*, char* buf, size_t cap) {
    snprintf(buf, cap, "emulation=1;");
    return 0;
}
}  // namespace ga
