// Non-templated accessors of the fft.Domain analogue (see ntt.hip.h).
#include "plonk.hip.h"
namespace ga {
void plonk_fixed_delete(PlonkFixed* fx) { plonk_fixed_destroy(fx); }
Domain* plonk_fixed_domain0(PlonkFixed* fx) { return fx->d0; }
void ntt_domain_delete(Domain* d) {
    if (!d) return;
    domain_free(d);
    delete d;
}
int ntt_domain_curve(const Domain* d) { return d->curve; }
uint64_t ntt_domain_size(const Domain* d) { return d->n; }
Ctx* ntt_domain_ctx(const Domain* d) { return d->ctx; }
}  // namespace ga
