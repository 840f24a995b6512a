// probabilistic_point_cloud_registration — command-line front end with the reference's flags, defaults
// and outputs (src/prob_point_cloud_registration_ex.cc:34-66,93-188, README.md:31-92):
//   probabilistic_point_cloud_registration [--dump] [-g gt.pcd] [-v] [-u] [-n int] [-c float] [-r float]
//       [-d float] [-i int] [-m int] [-t float] [-s float] <source.pcd> <target.pcd>
// Additions (do not collide with the reference's letters): --device N, --inner-steps K.
#include <cstdlib>
#include <filesystem>
#include <fstream>
#include <iostream>
#include <limits>
#include <memory>
#include <string>
#include <vector>

#include "prob_point_cloud_registration/pcd_io.hpp"
#include "prob_point_cloud_registration/prob_point_cloud_registration.h"
#include "prob_point_cloud_registration/utilities.hpp"

using prob_point_cloud_registration::ProbPointCloudRegistration;
using prob_point_cloud_registration::ProbPointCloudRegistrationParams;
typedef pcl::PointXYZ PointType;

namespace {

struct ArgError {
    std::string what, arg;
};

[[noreturn]] void usage_and_exit(const ArgError &e)
{
    std::cerr << "error: " << e.what << " for arg " << e.arg << std::endl;
    std::cerr << "usage: probabilistic_point_cloud_registration [--dump] [-g <string>] [-v] [-u] [-n <int>] [-c <float>]\n"
                 "         [-r <float>] [-d <float>] [-i <int>] [-m <int>] [-t <float>] [-s <float>] [--device <int>]\n"
                 "         [--inner-steps <int>] <source_file_name> <target_file_name>"
              << std::endl;
    std::exit(EXIT_FAILURE);
}

}  // namespace

int main(int argc, char **argv)
{
    bool use_gaussian = false, ground_truth = false;
    std::string source_file_name, target_file_name, ground_truth_file_name;
    ProbPointCloudRegistrationParams params;
    // CLI defaults (they differ from the struct's: radius 3 here, 1 there — kept as in the reference)
    params.max_neighbours = 20;
    params.n_iter = 1000;
    params.dof = 5;
    params.radius = 3;
    params.cost_drop_thresh = 0.01;
    params.n_cost_drop_it = 5;
    std::vector<std::string> positional;
    try {
        for (int i = 1; i < argc; i++) {
            const std::string a = argv[i];
            auto value = [&](const std::string &name) -> std::string {
                if (i + 1 >= argc) throw ArgError{"Missing a value", name};
                return argv[++i];
            };
            auto as_int = [&](const std::string &name) {
                const std::string v = value(name);
                std::size_t pos = 0;
                int r = 0;
                try {
                    r = std::stoi(v, &pos);
                } catch (...) {
                    throw ArgError{"Couldn't read argument value from string '" + v + "'", name};
                }
                if (pos != v.size()) throw ArgError{"Couldn't read argument value from string '" + v + "'", name};
                return r;
            };
            auto as_float = [&](const std::string &name) {
                const std::string v = value(name);
                std::size_t pos = 0;
                float r = 0;
                try {
                    r = std::stof(v, &pos);
                } catch (...) {
                    throw ArgError{"Couldn't read argument value from string '" + v + "'", name};
                }
                if (pos != v.size()) throw ArgError{"Couldn't read argument value from string '" + v + "'", name};
                return r;
            };
            if (a == "-s" || a == "--source_filter_size") params.source_filter_size = as_float(a);
            else if (a == "-t" || a == "--target_filter_size") params.target_filter_size = as_float(a);
            else if (a == "-m" || a == "--max_neighbours") params.max_neighbours = as_int(a);
            else if (a == "-i" || a == "--num_iter") params.n_iter = as_int(a);
            else if (a == "-d" || a == "--dof") params.dof = as_float(a);
            else if (a == "-r" || a == "--radius") params.radius = as_float(a);
            else if (a == "-c" || a == "--cost_drop_treshold") params.cost_drop_thresh = as_float(a);
            else if (a == "-n" || a == "--num_drop_iter") params.n_cost_drop_it = as_int(a);
            else if (a == "-u" || a == "--use_gaussian") use_gaussian = true;
            else if (a == "-v" || a == "--verbose") params.verbose = true;
            else if (a == "-g" || a == "--ground_truth") {
                ground_truth = true;
                ground_truth_file_name = value(a);
            } else if (a == "--dump") params.summary = true;
            else if (a == "--device") params.device_id = as_int(a);
            else if (a == "--inner-steps") params.inner_max_steps = as_int(a);
            else if (a == "-h" || a == "--help") throw ArgError{"help requested", a};
            else if (a.size() > 1 && a[0] == '-' && !(a[1] >= '0' && a[1] <= '9') && a[1] != '.') throw ArgError{"Couldn't find match for argument", a};
            else positional.push_back(a);
        }
        if (positional.size() < 1) throw ArgError{"Required argument missing", "source_file_name"};
        if (positional.size() < 2) throw ArgError{"Required argument missing", "target_file_name"};
        if (positional.size() > 2) throw ArgError{"Too many positional arguments", positional[2]};
    } catch (const ArgError &e) {
        usage_and_exit(e);
    }
    source_file_name = positional[0];
    target_file_name = positional[1];

    if (use_gaussian) {
        if (params.verbose) std::cout << "Using gaussian model" << std::endl;
        params.dof = std::numeric_limits<double>::infinity();
    } else if (params.verbose) {
        std::cout << "Using a t-distribution with " << params.dof << " dof" << std::endl;
    }
    if (params.verbose) {
        std::cout << "Radius of the neighborhood search: " << params.radius << std::endl;
        std::cout << "Max number of neighbours: " << params.max_neighbours << std::endl;
        std::cout << "Max number of iterations: " << params.n_iter << std::endl;
        std::cout << "Cost drop threshold: " << params.cost_drop_thresh << std::endl;
        std::cout << "Num cost drop iter: " << params.n_cost_drop_it << std::endl;
        std::cout << "Loading source point cloud from " << source_file_name << std::endl;
    }
    namespace pio = prob_point_cloud_registration::io;
    auto source_cloud = std::make_shared<pcl::PointCloud<PointType>>();
    if (pio::loadPCDFile(source_file_name, *source_cloud) == -1) {
        std::cout << "Could not load source cloud, closing" << std::endl;
        std::exit(EXIT_FAILURE);
    }
    if (params.verbose) std::cout << "Loading target point cloud from " << target_file_name << std::endl;
    auto target_cloud = std::make_shared<pcl::PointCloud<PointType>>();
    if (pio::loadPCDFile(target_file_name, *target_cloud) == -1) {
        std::cout << "Could not load target cloud, closing" << std::endl;
        std::exit(EXIT_FAILURE);
    }
    pcl::PointCloud<PointType>::Ptr source_ground_truth;
    if (ground_truth) {
        std::cout << "Loading ground truth point cloud from " << ground_truth_file_name << std::endl;
        source_ground_truth = std::make_shared<pcl::PointCloud<PointType>>();
        if (pio::loadPCDFile(ground_truth_file_name, *source_ground_truth) == -1) {
            std::cout << "Could not load ground truth" << std::endl;
            ground_truth = false;  // continue without it, like the reference
        }
    }

    std::unique_ptr<ProbPointCloudRegistration> registration;
    try {
        if (ground_truth)
            registration = std::make_unique<ProbPointCloudRegistration>(source_cloud, target_cloud, params, source_ground_truth);
        else
            registration = std::make_unique<ProbPointCloudRegistration>(source_cloud, target_cloud, params);
        if (params.verbose) std::cout << "Registration\n";
        registration->align();
    } catch (const std::exception &e) {
        std::cerr << "registration failed: " << e.what() << std::endl;
        return EXIT_FAILURE;
    }
    if (registration->transformation_history().empty()) {
        std::cerr << "no iteration was performed (num_iter = 0?)" << std::endl;
        return EXIT_FAILURE;
    }
    const Eigen::Affine3d estimated_transform = registration->transformation();
    auto aligned_source = std::make_shared<pcl::PointCloud<PointType>>();
    pcl::transformPointCloud(*source_cloud, *aligned_source, estimated_transform);
    if (params.verbose) {
        std::cout << "Transformation history:" << std::endl;
        for (const auto &trans : registration->transformation_history()) {
            const Eigen::Quaterniond rotq(trans.rotation());
            std::cout << "T: " << trans.translation().x() << ", " << trans.translation().y() << ", " << trans.translation().z()
                      << " ||| R: " << rotq.x() << ", " << rotq.y() << ", " << rotq.z() << ", " << rotq.w() << std::endl;
        }
        const std::filesystem::path source_path(source_file_name);
        const std::string aligned_source_name = "aligned_" + source_path.filename().string();
        std::cout << "Saving aligned source cloud to: " << aligned_source_name << std::endl;
        pio::savePCDFile(aligned_source_name, *aligned_source);  // only when verbose, as in the reference
    }
    if (params.summary) {
        const std::filesystem::path source_path(source_file_name), target_path(target_file_name);
        const std::string report_file_name = source_path.stem().string() + "_" + target_path.stem().string() + "_summary.txt";
        std::cout << "Saving registration report to: " << report_file_name << std::endl;
        std::ofstream report_file(report_file_name);
        report_file << "Source: " << source_file_name << " with filter size: " << params.source_filter_size << std::endl;
        report_file << "Target:" << target_file_name << " with filter size: " << params.target_filter_size << std::endl;
        report_file << "dof: " << params.dof << " | Radius: " << params.radius << " | Max_iter: " << params.n_iter
                    << " | Max neigh: " << params.max_neighbours << " | Cost_drop_thresh_: " << params.cost_drop_thresh
                    << " | N_cost_drop_it: " << params.n_cost_drop_it << std::endl;
        report_file << registration->report();
    }
    if (ground_truth) {
        const double mse_gtruth = prob_point_cloud_registration::calculateMSE(aligned_source, source_ground_truth);
        std::cout << "MSE w.r.t. ground truth: " << mse_gtruth << std::endl;
    }
    return 0;
}
