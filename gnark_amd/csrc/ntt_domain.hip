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
Domain* ntt_domain_take_spare(Ctx* ctx, int curve, uint64_t n) {
    std::lock_guard<std::mutex> g(ctx->spare_mu);
    Domain* d = ctx->spare_domain;
    if (!d || d->curve != curve || d->n != n) return nullptr;
    ctx->spare_domain = nullptr;
    return d;
}
void ntt_domain_give_spare(Ctx* ctx, Domain* d) {
    if (!d) return;
    const char* e = getenv("GA_DOMAIN_SPARE");
    Domain* old = nullptr;
    if (e && atoi(e) == 0) {
        old = d;
    } else {
        std::lock_guard<std::mutex> g(ctx->spare_mu);
        old = ctx->spare_domain;
        ctx->spare_domain = d;
    }
    ntt_domain_delete(old);
}
int ntt_domain_curve(const Domain* d) { return d->curve; }
uint64_t ntt_domain_size(const Domain* d) { return d->n; }
Ctx* ntt_domain_ctx(const Domain* d) { return d->ctx; }
}  // namespace ga
