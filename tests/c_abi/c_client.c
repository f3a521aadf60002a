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
        printf("gpu: world created\n");
        if (mgx_world_destroy(w) != MGX_OK) return 8;
    } else if (rc == MGX_ERR_NO_DEVICE) {
        printf("no gpu: %s\n", mgx_last_error());
    } else {
        return 9;
    }
    printf("c client ok\n");
    return 0;
}
