// mgx_world.hip — host side of the C ABI (include/mgx.h): world arena, topology -> slot tables,
// device buffers, launch sequencing.  No CPU compute path exists: every compute entry point
// needs a HIP device.
//
// Host keeps a per-robot / per-connection mirror of all mutable state only to (re)build the
// device arrays when the topology changes (`commit`): device is the source of truth between
// commits; `pull` downloads it first so that added robots / inter-robot connections never
// disturb existing state (the reference mutates its graphs in place,
// factorgraph.rs:190-226,304-353,380-436).
//
// ONE translation unit, ten files: the parts below share file-local helpers (commit, confirm_resident, flush_counts, sweep,
// run_resident, linger_close ...) that have no business in the library's symbol table, so they are included here in
// dependency order instead of being linked:
//   mgx_world_types.h — what the host side is made of: launcher prototypes of the kernel files, error text, device buffers, the pinned argument ring, the host mirror's records (Robot, IrConn), connection sets and index, the world itself, the entry hooks of the C ABI (MGX_ENTER)
//   mgx_world_mirror.inc — SoA helpers, blob <-> host mirror, pull (device -> host mirror)
//   mgx_world_counters.inc — message counters (MessageCount): the launch log, lazy settling of the connections, flush_counts
//   mgx_world_commit.inc — commit (host mirror -> device arrays), the connection index, the incoming tables, retopo (edge tables rebuilt on the device)
//   mgx_world_launch.inc — launches: confirm_resident, sweep, resident schedule launches, lingering launches (the host's side)
//   mgx_world_abi.inc — C ABI: lifecycle, environment, robots, connections, factor kinds, flags
//   mgx_world_topology.inc — C ABI: dynamic inter-robot topology — neighbour search, delete / create_interrobot_factors
//   mgx_world_missions.inc — C ABI: missions on the device, many ticks per call, fine-grained sweeps, schedules -> launches
//   mgx_world_schedule.inc — C ABI: batches, mgx_iterate, prior changes, mgx_tick, resets, diagnostics, read-back
//   mgx_world_shard.inc — C ABI: sharded worlds — exchange lists, migration, the in-engine transports (RCCL, direct, ghost records inside resident launches), hipIpc, pack / unpack
#include "mgx_world_types.h"
#include "mgx_world_mirror.inc"
#include "mgx_world_counters.inc"
#include "mgx_world_commit.inc"
#include "mgx_world_launch.inc"
#include "mgx_world_abi.inc"
#include "mgx_world_topology.inc"
#include "mgx_world_missions.inc"
#include "mgx_world_schedule.inc"
#include "mgx_world_shard.inc"
