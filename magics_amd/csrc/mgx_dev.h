// mgx_dev.h — device-side view of a world (plain pointers, passed to kernels by value).
//
// HBM layout: every per-item quantity is stored component-major ("SoA"):
//   arr[c * stride + item], so that lanes working on consecutive items issue coalesced
//   8-byte loads for each component c.  Items:
//     variables       v = robot * K + i                      (stride V, ghosts at the end)
//     internal edges  e = robot * E + slot, E = 4K - 6       (stride EI)
//         slot [0, K-1)        dynamic factor i   -> variable i       (factor slot 0)
//         slot [K-1, 2K-2)     dynamic factor i   -> variable i + 1   (factor slot 1)
//         slot [2K-2, 3K-4)    obstacle factor j  -> variable j + 1
//         slot [3K-4, 4K-6)    tracking factor j  -> variable j + 1
//     inter-robot edges, stored at their TARGET variable (CSR by variable, ir_var_ptr)
#pragma once
#include <cstdint>

namespace mgx {

constexpr int SNAP_W = 24;  // eta(4) lam(16) mu(4)

struct DevWorld {
    int R_local, R_total, K, E;
    int V, EI, ND, NT, NI;
    int cur;  // snapshot buffer read by this launch; the other one is written
    uint32_t enable;

    // variables
    double *prior_eta, *prior_lam;
    double *bel_eta, *bel_lam, *bel_mu, *bel_cov;
    int32_t *bel_valid;
    double *snap[2];          // [V][24] variable -> own-factor snapshot (eta, lam, mu), one 192-B record per
                              // variable: other robots' workgroups gather whole records
    uint32_t *snap_epoch[2];  // [V] number of deliveries (internal sweeps + prior changes)

    // internal factor -> variable messages
    double *fv_eta, *fv_lam;  // [4][EI], [16][EI]
    double *dyn_m;            // [16][ND] compact J^T Q J of each dynamic factor
    // tracking factor state
    int32_t *trk_record;      // [NT]
    float *trk_last_pos;      // [2][NT]
    double *trk_last_val;     // [NT]
    const int32_t *path_ptr;  // [R_local + 1]
    const float *path_xy;
    int32_t *iter_factor;     // [R_local] FactorGraph.iteration_count.factor

    // inter-robot edges (at the target variable)
    const int32_t *ir_var_ptr;    // [R_local * K + 1]
    const int32_t *ir_var_mid;    // [R_local * K] first edge whose owner has a HIGHER graph key
    const int32_t *ir_src_var;    // [NI] snapshot index of the owner's variable
    const int32_t *ir_dst_var;    // [NI] target variable
    const int32_t *ir_src_robot;  // [NI]
    const double *ir_dsafe, *ir_off;
    const uint8_t *ir_dst_slot;   // 1 if the target graph has the higher order key
    const uint32_t *ir_created;   // epoch of the owner's variable when the factor was created
    double *ir_fv_eta, *ir_fv_lam;  // [4][NI],[16][NI] factor -> target variable
    double *ir_bmu;                 // [4][NI] mean of the target variable -> factor message (the
                                    // only part of that inbox entry the kept output depends on)

    const uint8_t *antenna, *idle;  // [R_total]

    // obstacle image
    const uint8_t *sdf;
    uint32_t sdf_w, sdf_h;
    double world_w, world_h, obs_delta;
    int ir_max_edges;  // largest number of inter-robot edges attached to one robot (LDS staging size)

    double inv_s2_obs, inv_s2_ir, inv_s2_trk, trk_pad, trk_attr;
    // diagnostic builds only (-DMGX_STAMPS, tools/stamps.py): per-workgroup phase cycle sums
    unsigned long long *dbg;
};

// phases of one launch
constexpr uint32_t PH_EXT_FACTOR = 1u;    // external_factor_iteration (+ routing)
constexpr uint32_t PH_EXT_VARIABLE = 2u;  // external_variable_iteration (+ routing)
constexpr uint32_t PH_INT_FACTOR = 4u;    // internal_factor_iteration
constexpr uint32_t PH_INT_VARIABLE = 8u;  // internal_variable_iteration

}  // namespace mgx
