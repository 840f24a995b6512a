// Outer registration driver on top of the C ABI (include/ppcr.h).  Behaviour follows the reference's
// src/prob_point_cloud_registration.cc (what is filtered and copied at construction, what one outer iteration
// reports, when the loop stops, what is printed); the organisation is this library's own: a State object that
// owns the device handle, with the clouds resident on the GPU from construction to the end of align().
#include "prob_point_cloud_registration/prob_point_cloud_registration.h"

#include <cmath>
#include <cstdint>
#include <iostream>
#include <sstream>

#include "prob_point_cloud_registration/adapters.hpp"
#include "prob_point_cloud_registration/utilities.hpp"

namespace prob_point_cloud_registration {

namespace {

using Cloud = pcl::PointCloud<pcl::PointXYZ>;

const float *xyz_of(const Cloud &cloud) { return cloud.empty() ? nullptr : &cloud[0].x; }
float *xyz_of(Cloud &cloud) { return cloud.empty() ? nullptr : &cloud[0].x; }
int64_t count_of(const Cloud &cloud) { return static_cast<int64_t>(cloud.size()); }

// Ceres' function_tolerance as the reference sets it (cc:97 there: "10e-6")
constexpr double kFunctionTolerance = 10e-6;

const char *const kReportColumns =
    "iter, n_success_steps, initial_cost, final_cost, tx, ty, tz, roll, pitch, yaw, mse_prev_iter, mse_gtruth";

}  // namespace

// pcl::VoxelGrid centroid down-sampling on the device (ppcr_voxel_filter): one point per occupied voxel, voxels in
// ascending index.  (Parity with PCL itself is unpinned: the reference has no test for it and PCL is not
// available here; the definition is pinned against the CPU restatement in the test tree.)
void voxelGridFilter(const Cloud &in, double leaf, Cloud &out, int device_id)
{
    if (!(leaf > 0) || in.empty()) {
        if (&in != &out) out = in;
        return;
    }
    Cloud kept;
    kept.points.resize(in.size());
    int64_t n_kept = 0;
    const int rc = ppcr_voxel_filter(device_id, xyz_of(in), count_of(in), sizeof(pcl::PointXYZ), static_cast<float>(leaf),
                                     xyz_of(kept), sizeof(pcl::PointXYZ), &n_kept);
    if (rc != PPCR_OK) throw DeviceError(rc, std::string("ppcr_voxel_filter: ") + ppcr_last_error(nullptr));
    kept.points.resize(static_cast<size_t>(n_kept));
    out.points.swap(kept.points);
}

// Everything the driver remembers between calls.
struct ProbPointCloudRegistration::State {
    ProbPointCloudRegistrationParams params;
    OutputStream log;            // verbose-gated stdout
    DeviceContext gpu;           // one C-ABI handle: device + stream + resident clouds
    ppcr_stop_rule stop{0, 0, 0.0};
    Cloud::Ptr target;           // the caller's cloud (shared; filtered in place when asked)
    Cloud full_source;           // private copy of the caller's source (what was uploaded; the moved clouds stay on the device)
    bool has_companion = false;  // the device also carries the unfiltered source next to the filtered one
    bool has_truth = false;
    double truth_distance = 0;   // mean distance to the ground truth after the last iteration
    std::vector<Eigen::Affine3d> cumulative;  // T_cum after every outer iteration
    std::ostringstream table;    // --dump rows

    State(const ProbPointCloudRegistrationParams &p, const Cloud &source, Cloud::Ptr target_cloud)
        : params(p), log(p.verbose), gpu(p.device_id), target(std::move(target_cloud)), full_source(source)
    {
    }

    ppcr_ctx *handle() const { return gpu.get(); }

    // the two farewell lines of the loop (verbose only), keyed by ppcr_stop_rule_check's verdict
    void explainStop(int verdict)
    {
        if (verdict == PPCR_STOP_MAX_ITERATIONS)
            log << "Terminating because maximum number of iterations has been reached ( " << stop.iteration << " iter)\n";
        else if (verdict == PPCR_STOP_COST_DROP)
            log << "Terminating because cost drop has been under " << params.cost_drop_thresh * 100 << " % for more than "
                << params.n_cost_drop_it << " iterations\n";
    }

    void uploadClouds()
    {
        // association cloud = the down-sampled source when a leaf size was given; the full-resolution copy then
        // rides along on the device as the handle's companion (moved by every iteration, read by the reports)
        Cloud thinned;
        const Cloud *assoc = &full_source;
        if (params.source_filter_size > 0) {
            log << "Filtering source point cloud with leaf of size " << params.source_filter_size << "\n";
            voxelGridFilter(full_source, params.source_filter_size, thinned, params.device_id);
            assoc = &thinned;
            has_companion = true;
        }
        if (params.target_filter_size > 0) {
            log << "Filtering target point cloud with leaf of size " << params.target_filter_size << "\n";
            voxelGridFilter(*target, params.target_filter_size, *target, params.device_id);
        }
        gpu.check(ppcr_set_params(handle(), params.radius, params.max_neighbours, params.dof, DIMENSIONS), "ppcr_set_params");
        gpu.check(ppcr_set_target(handle(), xyz_of(*target), count_of(*target), sizeof(pcl::PointXYZ)), "ppcr_set_target");
        gpu.check(ppcr_set_source(handle(), xyz_of(*assoc), count_of(*assoc), sizeof(pcl::PointXYZ)), "ppcr_set_source");
        if (has_companion)
            gpu.check(ppcr_set_companion(handle(), xyz_of(full_source), count_of(full_source), sizeof(pcl::PointXYZ)),
                      "ppcr_set_companion");
        if (params.summary) table << kReportColumns << std::endl;
    }

    void attachGroundTruth(const Cloud &truth)
    {
        gpu.check(ppcr_set_ground_truth(handle(), xyz_of(truth), count_of(truth), sizeof(pcl::PointXYZ)),
                  "ppcr_set_ground_truth");
        has_truth = true;
        measureTruthDistance();
        log << "Initial MSE w.r.t. ground truth: " << truth_distance << "\n";
    }

    void measureTruthDistance() { gpu.check(ppcr_mse_ground_truth(handle(), &truth_distance), "ppcr_mse_ground_truth"); }

    // The whole loop of align() as ONE call: the device paces itself (association, inner loop to function_tolerance,
    // solve, both source moves, the two mean distances) and runs an iteration ahead of this thread; what the reference
    // prints and tabulates between two iterations (cc:114-129 there) arrives through absorb(), in order, when that
    // iteration's numbers reach the host.
    void runLoop()
    {
        const int reports = (has_truth ? PPCR_REPORT_TRUTH : 0) | (params.summary ? PPCR_REPORT_MOVED : 0);
        gpu.check(ppcr_align_report(handle(), params.n_iter, params.cost_drop_thresh, params.n_cost_drop_it,
                                    params.initial_rotation, params.initial_translation, params.inner_max_steps,
                                    kFunctionTolerance, &stop, reports, &State::onIteration, this, nullptr, nullptr),
                  "ppcr_align_report");
    }

    static void onIteration(void *self, const ppcr_iteration_info *info) { static_cast<State *>(self)->absorb(*info); }

    void absorb(const ppcr_iteration_info &it)
    {
        const Eigen::Affine3d delta = affineFromRows(it.T_step);
        cumulative.push_back(cumulative.empty() ? delta : delta * cumulative.back());
        log << "iteration " << it.iteration << ": initial_cost " << it.cost[0] << " final_cost " << it.cost[1] << " inner steps "
            << it.inner_steps << "\n";
        if (has_truth) {
            truth_distance = it.mse_truth;
            log << "MSE w.r.t. ground truth: " << truth_distance << "\n";
        }
        if (params.summary) appendRow(it);
    }

    void appendRow(const ppcr_iteration_info &it)
    {
        const Eigen::Affine3d &T = cumulative.back();
        const Eigen::Vector3d angles = T.rotation().eulerAngles(0, 1, 2);
        table << it.iteration << ", " << it.inner_steps << ", " << it.cost[0] << ", " << it.cost[1];
        for (int a = 0; a < 3; a++) table << ", " << T.translation()(a);
        for (int a = 0; a < 3; a++) table << ", " << pcl::rad2deg(angles(a));
        table << ", " << it.moved << ", " << truth_distance << std::endl;  // it.moved: mean distance each point travelled
    }
};

ProbPointCloudRegistration::ProbPointCloudRegistration(Cloud::Ptr source_cloud, Cloud::Ptr target_cloud,
                                                       ProbPointCloudRegistrationParams parameters)
    : state_(new State(parameters, *source_cloud, std::move(target_cloud)))
{
    state_->uploadClouds();
}

ProbPointCloudRegistration::ProbPointCloudRegistration(Cloud::Ptr source_cloud, Cloud::Ptr target_cloud,
                                                       ProbPointCloudRegistrationParams parameters,
                                                       Cloud::Ptr ground_truth_cloud)
    : ProbPointCloudRegistration(std::move(source_cloud), std::move(target_cloud), parameters)
{
    state_->attachGroundTruth(*ground_truth_cloud);
}

ProbPointCloudRegistration::~ProbPointCloudRegistration() = default;

void ProbPointCloudRegistration::align()
{
    state_->runLoop();    // while (!hasConverged()) { one outer iteration }, on the device
    (void)hasConverged();  // the loop's last look at the rule: prints why it stopped (verbose)
    if (state_->has_truth) {
        state_->measureTruthDistance();
        std::cout << "MSE w.r.t. ground truth: " << state_->truth_distance << std::endl;
    }
}

bool ProbPointCloudRegistration::hasConverged()
{
    State &st = *state_;
    const int verdict = ppcr_stop_rule_check(&st.stop, st.params.n_iter, st.params.cost_drop_thresh, st.params.n_cost_drop_it);
    st.explainStop(verdict);
    return verdict != PPCR_CONTINUE;
}

Eigen::Affine3d ProbPointCloudRegistration::transformation()
{
    return state_->cumulative.empty() ? Eigen::Affine3d::Identity() : state_->cumulative.back();
}

std::vector<Eigen::Affine3d> ProbPointCloudRegistration::transformation_history() { return state_->cumulative; }

std::string ProbPointCloudRegistration::report() { return state_->table.str(); }

}  // namespace prob_point_cloud_registration
