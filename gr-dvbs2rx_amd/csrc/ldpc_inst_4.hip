// one kernel variant per translation unit (parallel build); see ldpc_kernel.hpp
#define DVBS2_LDPC_INSTANTIATE 4
#include "ldpc_kernel.hpp"
