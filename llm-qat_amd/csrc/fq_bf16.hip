// fq_bf16.hip -- kernel instantiations and launch logic for BF16 tensors.
#include "fq_dtype_impl.h"
namespace fq {
FQ_INSTANTIATE(BF16)
}
