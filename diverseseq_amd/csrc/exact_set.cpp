// Host tie arbiter -- placeholder until the exact replay lands.
#include "select.h"

int dvs_select_arbitrate(dvs_ctx *ctx, dvs_select *s) {
    const SelCtl &c = *s->h_ctl;
    return dvs_set_error(ctx, DVS_ERR_UNSUPPORTED,
                         "ambiguous decision at stream position %llu (stage %u) and no arbiter built",
                         (unsigned long long)c.arb_pos, c.arb_stage);
}
void dvs_select_arbiter_free(dvs_select *) {}
