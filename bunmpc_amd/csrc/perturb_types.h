// Launch descriptor of the perturbation sampler (perturb.hip), filled by bmpc_perturb_batch_device.
#pragma once
#include "../../include/bunmpc.h"
#include "ik_types.h"

namespace bunmpc {

struct PerturbLaunch {
    bmpc_perturb_batch_t d;
    const RobotModelDev *model;
};

int launch_perturb(const PerturbLaunch &a, hipStream_t st);

}  // namespace bunmpc
