// TEMPORARY stub, replaced by the MFMA networks.
#include "fp_nn.h"
namespace fp {
struct Net {}; struct NNScratch {};
Net *net_load(const char *, bool, std::string *err) { if (err) *err = "NN not built yet"; return nullptr; }
void net_free(Net *n) { delete n; }
NNScratch *nn_scratch_create() { return new NNScratch(); }
void nn_scratch_free(NNScratch *w) { delete w; }
int refiner_forward(hipStream_t, Profiler *, const Net *, NNScratch *, const __half *, int, float *, float *) { set_error("NN not built"); return 1; }
int scorer_features(hipStream_t, Profiler *, const Net *, NNScratch *, const __half *, int, float *) { set_error("NN not built"); return 1; }
int scorer_head(hipStream_t, Profiler *, const Net *, NNScratch *, const float *, int, float *) { set_error("NN not built"); return 1; }
}
