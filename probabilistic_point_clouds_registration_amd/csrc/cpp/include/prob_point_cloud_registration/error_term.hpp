// ErrorTerm — the per-correspondence residual functor of the reference API (error_term.hpp:10-51):
// r = y - (R(q) x + t), q = (w,x,y,z) normalised before use, templated on the scalar so user code that
// evaluates residuals (or differentiates them with its own Jet type) keeps working.  In this
// implementation the registration itself never instantiates ErrorTerms: the same residual is evaluated
// for every stored pair inside the HIP kernels (sq_residual in ppcr_device.hip.h).
#pragma once
#include <cmath>
#include <memory>

#include "prob_point_cloud_registration/compat.hpp"

namespace prob_point_cloud_registration {

class ErrorTerm {
public:
    static const int kResiduals = 3;
    ErrorTerm(const pcl::PointXYZ source_point, const pcl::PointXYZ target_point)
        : source_{source_point.x, source_point.y, source_point.z}, target_{target_point.x, target_point.y, target_point.z},
          // rho(s) = w s with w = 1 to begin with, swappable while a problem holds the wrapper (error_term.hpp:17-19)
          weight_(std::make_shared<Held>(new ceres::LossFunctionWrapper(new ceres::ScaledLoss(NULL, 1.0, ceres::TAKE_OWNERSHIP), ceres::TAKE_OWNERSHIP)))
    {
    }

    template <typename T>
    bool operator()(const T *const rotation, const T *const translation, T *residuals) const
    {
        using std::sqrt;
        const T n = sqrt(rotation[0] * rotation[0] + rotation[1] * rotation[1] + rotation[2] * rotation[2] +
                         rotation[3] * rotation[3]);
        const T w = rotation[0] / n, x = rotation[1] / n, y = rotation[2] / n, z = rotation[3] / n;
        const T p[3] = {T(source_[0]), T(source_[1]), T(source_[2])};
        // v' = v + 2 w (u x v) + 2 u x (u x v), u = (x,y,z)
        const T c0 = y * p[2] - z * p[1], c1 = z * p[0] - x * p[2], c2 = x * p[1] - y * p[0];
        const T d0 = y * c2 - z * c1, d1 = z * c0 - x * c2, d2 = x * c1 - y * c0;
        const T rot[3] = {p[0] + T(2) * (w * c0 + d0), p[1] + T(2) * (w * c1 + d1), p[2] + T(2) * (w * c2 + d2)};
        for (int i = 0; i < kResiduals; i++) residuals[i] = T(target_[i]) - (rot[i] + translation[i]);
        return true;
    }

    // error_term.hpp:39-43: a new ScaledLoss(NULL, w) behind the same wrapper
    void updateWeight(double new_weight) { weight_->wrapper->Reset(new ceres::ScaledLoss(NULL, new_weight, ceres::TAKE_OWNERSHIP), ceres::TAKE_OWNERSHIP); }
    // error_term.hpp:45: the wrapper a caller hands to ceres::Problem::AddResidualBlock (this library never builds a Ceres
    // problem itself; with the real Ceres — PPCR_NO_COMPAT_TYPES — this IS ceres::LossFunctionWrapper).
    // OWNERSHIP AS IN THE REFERENCE: the reference's term holds a raw pointer and never deletes it — the ceres::Problem the
    // wrapper is added to does (Problem::Options::loss_function_ownership defaults to TAKE_OWNERSHIP, ..._iteration.hpp:42-44).
    // So the wrapper is the CALLER'S from the first time this accessor hands it out: the term (and its copies, which
    // share the one wrapper as the reference's copies do) only deletes a wrapper nobody ever asked for.  A term whose
    // weight() was taken and then added to no problem leaks it, exactly as in the reference.
    ceres::LossFunctionWrapper *weight()
    {
        weight_->handed_out = true;
        return weight_->wrapper;
    }
    // addition of this implementation: look at the weight without taking the wrapper over
    const ceres::LossFunctionWrapper *weight() const { return weight_->wrapper; }

    // additions of this implementation: the two points as the functor holds them (float values widened to double,
    // error_term.hpp:15-16) — what WeightUpdaterCallback uploads for its device route
    const double *source() const { return source_; }
    const double *target() const { return target_; }

private:
    // the wrapper and whether somebody took it over (shared by the copies of a term)
    struct Held {
        explicit Held(ceres::LossFunctionWrapper *w) : wrapper(w) {}
        ~Held()
        {
            if (!handed_out) delete wrapper;
        }
        Held(const Held &) = delete;
        Held &operator=(const Held &) = delete;
        ceres::LossFunctionWrapper *wrapper;
        bool handed_out = false;
    };
    double source_[3];
    double target_[3];
    std::shared_ptr<Held> weight_;
};

}  // namespace prob_point_cloud_registration
