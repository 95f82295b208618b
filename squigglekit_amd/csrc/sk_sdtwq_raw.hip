// sk_sdtwq_raw.hip -- the screening kernels (sk_sdtwq.hip) for already-normalised float64 samples
// (SK_FEED_F64_RAW, the mlpy boundary); a translation unit of its own so that the three feeds build in parallel.
#define SK_SDTWQ_FEED 2
#include "sk_sdtwq.hip"
