/* A plain C99 client of include/mgx.h (tests/test_abi.py): the boundary has to be consumable from C
 * — the binding language of cgo / bindgen / JNI alike.  Uses host-only entry points, so it runs
 * without a GPU; on a machine with one it also creates and destroys a world. */
#include <stdio.h>
#include <string.h>

#include "mgx.h"

int main(void) {
    uint8_t steps[64];
    uint32_t ts[64];
    double v[2] = {3.0, -4.0};
    double prec[4] = {2.0, 0.0, 0.0, 4.0}, info[2] = {1.0, 1.0}, mean[2];
    mgx_mvn *n = NULL;
    mgx_params p;
    mgx_world *w = NULL;
    int rc;

    if (mgx_schedule(MGX_SCHEDULE_INTERLEAVE_EVENLY, 10, 10, steps, 64) != 10) return 1;
    if (mgx_variable_timesteps(45, 3, ts, 64) != 16) return 2;
    if (mgx_euclidean_norm(v, 2) != 5.0 || mgx_l1_norm(v, 2) != 7.0) return 3;
    if (mgx_mvn_from_information_and_precision(info, 2, prec, 2, 2, &n) != MGX_OK) return 4;
    if (mgx_mvn_get(n, NULL, NULL, mean) != MGX_OK || mean[0] != 2.0 || mean[1] != 4.0) return 5; /* precision . information */
    mgx_mvn_destroy(n);
    if (mgx_mvn_from_information_and_precision(info, 2, prec, 2, 3, &n) != MGX_MVN_ERR_NON_SQUARE) return 6;
    if (strstr(mgx_last_error(), "NonSquarePrecisionMatrix(2, 3)") == NULL) return 7;

    memset(&p, 0, sizeof p);
    p.sigma_dynamics = 0.1; p.sigma_interrobot = 0.01; p.sigma_obstacle = 0.01; p.sigma_tracking = 0.01;
    p.safety_multiplier = 2.5; p.enable_mask = 7;
    rc = mgx_world_create(&p, &w);
    if (rc == MGX_OK) {
        /* the environment of config/scenarios/Junction Twoway as plain data: one crossroads tile */
        static const uint32_t tiles[1] = {0x253C};
        static uint8_t rgb[3 * 200 * 200];
        mgx_env_desc env;
        uint32_t iw = 0, ih = 0;
        memset(&env, 0, sizeof env);
        env.n_rows = 1; env.n_cols = 1; env.tiles = tiles; env.tile_size = 100.0f; env.path_width = 0.16f;
        env.sdf_resolution = 200; env.sdf_expansion = 0.01f; env.sdf_blur = 0.01f;
        printf("gpu: world created\n");
        if (mgx_env_image_size(&env, 200, &iw, &ih) != MGX_OK || iw != 200 || ih != 200) return 10;
        if (mgx_env_to_sdf_image(&env, 200, 0.01f, 0.01f, rgb) != MGX_OK) return 11;
        if (rgb[3 * (100 * 200 + 100)] != 255 || rgb[3 * (5 * 200 + 5)] != 0) return 12; /* road centre / corner block */
        if (mgx_world_set_environment(w, &env) != MGX_OK) return 13;
        env.path_width = 0.001f; /* path_width - expansion < 0: Percentage::new panics in the reference */
        if (mgx_env_to_sdf_image(&env, 200, 0.01f, 0.01f, rgb) != MGX_ERR_INVALID) return 14;
        if (mgx_set_enabled(w, MGX_FACTOR_DYNAMIC | MGX_FACTOR_OBSTACLE) != MGX_OK) return 15;
        if (mgx_world_destroy(w) != MGX_OK) return 8;
    } else if (rc == MGX_ERR_NO_DEVICE) {
        printf("no gpu: %s\n", mgx_last_error());
    } else {
        return 9;
    }
    printf("c client ok\n");
    return 0;
}
