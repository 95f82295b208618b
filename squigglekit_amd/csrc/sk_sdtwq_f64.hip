// sk_sdtwq_f64.hip -- the screening kernels (sk_sdtwq.hip) for float64 samples normalised on the fly
// (SK_FEED_F64_NORM); a translation unit of its own so that the three feeds build in parallel.
#define SK_SDTWQ_FEED 1
#include "sk_sdtwq.hip"
