// parity-in-records kernel variant (see ldpc_kernel_pr.hpp)
#define DVBS2_LDPC_INSTANTIATE_PR 1
#include "ldpc_kernel_pr.hpp"
