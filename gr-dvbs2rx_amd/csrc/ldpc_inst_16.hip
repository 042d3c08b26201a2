// one kernel variant per translation unit (parallel build); see ldpc_kernel.hpp
#define DVBS2_LDPC_INSTANTIATE 16
#include "ldpc_kernel.hpp"
