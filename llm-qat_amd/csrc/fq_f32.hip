// fq_f32.hip -- kernel instantiations and launch logic for F32 tensors.
#include "fq_dtype_impl.h"
namespace fq {
FQ_INSTANTIATE(F32)
}
