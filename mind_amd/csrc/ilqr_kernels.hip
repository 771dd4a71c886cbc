// placeholder: replaced by the tree-iLQR kernels
#include <hip/hip_runtime.h>
