// fq_f16.hip -- kernel instantiations and launch logic for F16 tensors.
#include "fq_dtype_impl.h"
namespace fq {
FQ_INSTANTIATE(F16)
}
