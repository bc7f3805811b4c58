// Launch descriptor of the inverse-dynamics output stage (id_ctrl.hip), filled by bmpc_id_batch_device.
#pragma once
#include "../../include/bunmpc.h"
#include "ik_types.h"

namespace bunmpc {

struct IdLaunch {
    bmpc_id_batch_t d;
    const RobotModelDev *model;
    int leg_foot[4];             // end-effector slot whose frame hangs off leg L (-1 = none)
    int leg_foot_k[4];           // ... on which body of the leg (0..2)
    double leg_foot_p[4][3];     // ... at which point of that body
};

int launch_id_batch(const IdLaunch &a, hipStream_t st);

}  // namespace bunmpc
