"""Python mirror of the reference's class surface over the C ABI (thin: every number comes from the HIP path).

    ProbPointCloudRegistrationParams   <- prob_point_cloud_registration_params.hpp:5-18
    ProbabilisticWeights               <- probabilistic_weights.hpp:18-106
    ProbPointCloudRegistrationIteration<- prob_point_cloud_registration_iteration.hpp:21-78
    ProbPointCloudRegistration         <- prob_point_cloud_registration.h:18-64 / src/...cc:15-158
"""
from dataclasses import dataclass, field

import numpy as np

from . import _lib


@dataclass
class ProbPointCloudRegistrationParams:
    max_neighbours: int = 20
    dof: float = 5.0
    radius: float = 1.0
    n_iter: int = 1000
    cost_drop_thresh: float = 0.01
    n_cost_drop_it: float = 5
    verbose: bool = False
    summary: bool = False
    initial_rotation: tuple = (1.0, 0.0, 0.0, 0.0)      # (w, x, y, z)
    initial_translation: tuple = (0.0, 0.0, 0.0)
    source_filter_size: float = 0.0                      # voxel-grid leaf sizes (0 = no filter), src/...cc:24-41
    target_filter_size: float = 0.0
    device_id: int = 0
    inner_max_steps: int = 100


class ProbabilisticWeights:
    def __init__(self, v, dimension, max_neighbours, device_id=0):
        if not (v > 0) or dimension <= 0:
            raise ValueError("v and dimension must be > 0")
        self.v, self.dimension, self.max_neighbours, self.device_id = float(v), int(dimension), int(max_neighbours), device_id

    def update_weights(self, row_ptr, squared_errors):
        """CSR pattern (row_ptr) + squared errors in storage order -> weights in the same order."""
        return _lib.update_weights(row_ptr, squared_errors, self.v, self.dimension, self.device_id)


class ProbPointCloudRegistrationIteration:
    def __init__(self, source_cloud, target_cloud, row_ptr, col, parameters):
        self.parameters = parameters
        self.ctx = _lib.Context(parameters.device_id)
        self.ctx.set_params(parameters.radius, parameters.max_neighbours, parameters.dof, 3)
        self.ctx.set_target(target_cloud)
        self.ctx.set_source(source_cloud)
        self.ctx.set_association(row_ptr, col)
        self._T = None

    def solve(self, function_tolerance=1e-6, max_num_iterations=50):
        T, cost, steps = self.ctx.solve(self.parameters.initial_rotation, self.parameters.initial_translation,
                                        max_steps=min(int(max_num_iterations), 100000), f_tol=function_tolerance)
        self._T = T
        return dict(initial_cost=cost[0], final_cost=cost[1], num_successful_steps=steps)

    def transformation(self):
        T = np.eye(4)
        if self._T is not None:
            T[:3] = self._T
        return T


class ProbPointCloudRegistration:
    def __init__(self, source_cloud, target_cloud, parameters, ground_truth_cloud=None):
        self.parameters = parameters
        source_cloud = np.ascontiguousarray(np.asarray(source_cloud, np.float32)[:, :3])
        target_cloud = np.ascontiguousarray(np.asarray(target_cloud, np.float32)[:, :3])
        self.ctx = _lib.Context(parameters.device_id)
        self.ctx.set_params(parameters.radius, parameters.max_neighbours, parameters.dof, 3)
        # the constructor's two pcl::VoxelGrid filters (src/...cc:24-41) run on the device; the association uses the
        # filtered source while the full one rides along as the handle's companion (cc:110-112)
        self.filtered = parameters.source_filter_size > 0
        if parameters.target_filter_size > 0:
            target_cloud = _lib.voxel_filter(target_cloud, parameters.target_filter_size, parameters.device_id)
        self.target_cloud = target_cloud
        self.ctx.set_target(target_cloud)
        if self.filtered:
            self.ctx.set_source(_lib.voxel_filter(source_cloud, parameters.source_filter_size, parameters.device_id))
            self.ctx.set_companion(source_cloud)
        else:
            self.ctx.set_source(source_cloud)
        self._history = []
        self._costs = []
        self.ground_truth = None
        if ground_truth_cloud is not None:
            self.ground_truth = np.ascontiguousarray(np.asarray(ground_truth_cloud, np.float32)[:, :3])
            self.ctx.set_ground_truth(self.ground_truth)

    def align(self):
        p = self.parameters
        res = self.ctx.align(p.n_iter, cost_drop_thresh=p.cost_drop_thresh, n_cost_drop_it=p.n_cost_drop_it,
                             q0=p.initial_rotation, t0=p.initial_translation, inner_steps=p.inner_max_steps, f_tol=10e-6)
        self._history = [np.vstack([T, [0, 0, 0, 1]]) for T in res["history"]]
        self._costs = res["costs"]
        return res["n_iter"]

    def transformation(self):
        return self._history[-1]

    def transformation_history(self):
        return list(self._history)

    def mse_ground_truth(self):
        """calculateMSE(source, ground truth): mean Euclidean distance of index-paired points (utilities.hpp:16-26)."""
        if self.ground_truth is None:
            return None
        return self.ctx.mse_ground_truth()      # device reduction over the full-resolution source

    def source_cloud(self):
        """The (moved) full-resolution source, as the reference's source_cloud_ after align()."""
        return self.ctx.get_companion() if self.filtered else self.ctx.get_source()
