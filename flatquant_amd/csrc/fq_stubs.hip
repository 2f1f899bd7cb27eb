// temporary: launchers not written yet report "unsupported"
#include "fq_common.hpp"
int fq_launch_block(int, const f16*, const f16*, int64_t, int, int, int, const FqQuantOut&, int, hipStream_t) { return -1000; }
int fq_launch_hadamard(const f16*, f16*, int64_t, int, int, const f16*, float, int, hipStream_t) { return -1000; }
