// Parameter block of the registration (same fields, order and defaults as the reference's
// prob_point_cloud_registration_params.hpp:5-18, so aggregate use and member access port unchanged).
#pragma once

namespace prob_point_cloud_registration {

struct ProbPointCloudRegistrationParams {
    int max_neighbours = 20;
    double dof = 5;
    double radius = 1;
    int n_iter = 1000;
    double cost_drop_thresh = 0.01;
    double n_cost_drop_it = 5;  // a double in the reference too
    bool verbose = false;
    bool summary = false;
    double initial_rotation[4] = {1, 0, 0, 0};  // (w, x, y, z)
    double initial_translation[3] = {0, 0, 0};
    double source_filter_size = 0;
    double target_filter_size = 0;
    // --- additions of this implementation (defaults reproduce the reference's behaviour) ---
    int device_id = 0;              // which GPU
    int inner_max_steps = 100;      // IRLS steps per association (the reference iterates LM to function_tolerance)
};

}  // namespace prob_point_cloud_registration
