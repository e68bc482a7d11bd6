/* A non-PyTorch, non-C++ host of the C ABI: compiled as C99 with -pedantic against include/r2l_hip.h by
 * tests/test_host_cpu.py::test_c99_host_compiles_links_and_runs and run WITHOUT a GPU — it only calls entry points that do no
 * device work (size queries, dispatch queries with an r2l_config, and the argument checks that fail before any launch). */
#include <stdio.h>
#include <string.h>

#include "r2l_hip.h"

#define CHECK(cond)                                                   \
    do {                                                              \
        if (!(cond)) {                                                \
            printf("FAILED line %d: %s\n", __LINE__, #cond);          \
            return 1;                                                 \
        }                                                             \
    } while (0)

int main(void) {
    r2l_config cfg;
    float dummy[4] = {0.f, 0.f, 0.f, 0.f};
    memset(&cfg, 0, sizeof cfg);
    CHECK(sizeof(r2l_config) == 32);
    CHECK(r2l_param_count(43) == 5917187);
    CHECK(r2l_teacher_param_count() == 595844);
    CHECK(r2l_padded_rows(4097) == 4128 && r2l_num_tiles(4097) == 129);
    CHECK(r2l_stash_slot_floats(32) > 0 && r2l_dw_slab_floats() > 0);
    CHECK(r2l_fwd_stream_floats(43) > 0 && r2l_bwd_stream_floats(43) > 0 && r2l_teacher_stream_floats() > 0);
    /* explicit dispatch: the bf16x3 family pinned, whatever the environment says */
    cfg.precision = R2L_PRECISION_BF16X3;
    CHECK(r2l_forward_layout_for_cfg(98304, 1, &cfg) == 3 && r2l_backward_layout_for_cfg(98304, &cfg) == 3);
    cfg.precision = R2L_PRECISION_AUTO;
    cfg.tiling = R2L_TILING_COOPF;
    cfg.coop_tiles = 2;
    CHECK(r2l_coop_tiles_for_cfg(4096, 43, &cfg) == 2);
    /* a bad config and bad arguments are error codes, with a message for this thread */
    cfg.dw_mode = 7;
    CHECK(r2l_variant_for_cfg(4096, &cfg) == -1);
    CHECK(strstr(r2l_last_error(), "dw_mode") != NULL);
    cfg.dw_mode = R2L_DW_EXACT;
    CHECK(r2l_forward_rays_cfg(NULL, dummy, NULL, dummy, dummy, dummy, 43, dummy, NULL, NULL, 32, NULL, &cfg) != 0);
    CHECK(strstr(r2l_last_error(), "r2l_forward_rays") != NULL);
    CHECK(r2l_forward_rays(NULL, NULL, NULL, NULL, NULL, NULL, 43, NULL, NULL, NULL, 0, NULL) == 0); /* N = 0: no-op */
    CHECK(r2l_adam_step(dummy, dummy, dummy, dummy, 4, 1e-3f, 0.9f, 0.999f, 1e-8f, 0, 1.f, NULL) != 0); /* step from 1 */
    CHECK(r2l_backward_part(dummy, dummy, NULL, dummy, NULL, dummy, dummy, NULL, dummy, dummy, dummy, dummy, 43, 1e-5f, dummy,
                            dummy, dummy, dummy, dummy, NULL, 32, NULL, 0, 0, 86) != 0); /* parts = 0 */
    printf("C99 host ok\n");
    return 0;
}
