/* explicit instantiations, see kmr_instances.hpp */
#include <hip/hip_runtime.h>
#define KMR_INSTANCE_TU      /* the plain (non-template) kernels of the headers belong to kmr_api.hip */
#define KMR_INST_PART
#define KMR_INST_W 4
#include "kmr_instances.hpp"
