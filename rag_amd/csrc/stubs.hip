// Entry points whose kernels are not built yet answer RAG_ERR_UNSUPPORTED.
#include "index.h"
namespace rag {
int ivf_assign_new_rows(rag_index*, long long, long long) { return fail(RAG_ERR_UNSUPPORTED, "IVF not built"); }
int ivf_search_device(rag_index*, long long, const float*, int, float, int, const uint8_t*, float*, long long*, hipStream_t) { return fail(RAG_ERR_UNSUPPORTED, "IVF not built"); }
void ivf_free(rag_index*) {}
}
using namespace rag;
extern "C" {
int rag_index_train_ivf(rag_index_t*, int, const float*, int64_t, const float*) { return fail(RAG_ERR_UNSUPPORTED, "IVF not built"); }
}
