// Minimal stand-ins for the third-party types on the reference's public surface (PCL point clouds,
// Eigen Affine3d / row-major sparse matrix, ceres::Solver::Options/Summary, the loss-function wrapper).  None of PCL,
// Eigen or Ceres is needed to build or use this library; the names, members and semantics below are a SUBSET of the real
// libraries' — every member a header of this directory touches exists under that name in the real library (what is
// specific to this implementation lives in adapters.hpp as free functions written against those members only).
// With the real libraries present, define PPCR_NO_COMPAT_TYPES: this header then includes them instead.
// tests/test_cpp_headers.py compiles every header and source of this directory against an API-subset mock of the three
// (tests/cpp/mock_real: real member names only), so a stand-in-only member cannot creep back in.
#pragma once
#include <cmath>
#include <cstddef>
#include <limits>
#include <memory>
#include <string>
#include <vector>

#ifdef PPCR_NO_COMPAT_TYPES

#include <Eigen/Geometry>
#include <Eigen/Sparse>
#include <ceres/ceres.h>
#include <pcl/common/angles.h>
#include <pcl/common/transforms.h>
#include <pcl/point_cloud.h>
#include <pcl/point_types.h>

#else

namespace pcl {

struct alignas(16) PointXYZ {  // same 16-byte layout as pcl::PointXYZ
    float x = 0, y = 0, z = 0, pad = 1.0f;
    PointXYZ() = default;
    PointXYZ(float x_, float y_, float z_) : x(x_), y(y_), z(z_) {}
};

template <class PointT>
class PointCloud {
public:
    using Ptr = std::shared_ptr<PointCloud<PointT>>;
    using ConstPtr = std::shared_ptr<const PointCloud<PointT>>;
    std::vector<PointT> points;
    std::size_t size() const { return points.size(); }
    bool empty() const { return points.empty(); }
    void reserve(std::size_t n) { points.reserve(n); }
    void resize(std::size_t n) { points.resize(n); }
    void clear() { points.clear(); }
    void push_back(const PointT &p) { points.push_back(p); }
    PointT &operator[](std::size_t i) { return points[i]; }
    const PointT &operator[](std::size_t i) const { return points[i]; }
    PointT &at(std::size_t i) { return points.at(i); }
    const PointT &at(std::size_t i) const { return points.at(i); }
    auto begin() { return points.begin(); }
    auto end() { return points.end(); }
    auto begin() const { return points.begin(); }
    auto end() const { return points.end(); }
};

inline double rad2deg(double r) { return r * 57.29577951308232; }

}  // namespace pcl

namespace Eigen {

struct Vector3d {
    double v[3] = {0, 0, 0};
    Vector3d() = default;
    Vector3d(double a, double b, double c) : v{a, b, c} {}
    double x() const { return v[0]; }
    double y() const { return v[1]; }
    double z() const { return v[2]; }
    double &operator()(int i) { return v[i]; }
    double operator()(int i) const { return v[i]; }
    double operator()(int i, int) const { return v[i]; }
    static Vector3d UnitX() { return Vector3d(1, 0, 0); }
    static Vector3d UnitY() { return Vector3d(0, 1, 0); }
    static Vector3d UnitZ() { return Vector3d(0, 0, 1); }
};

struct Matrix3d {
    double m[3][3] = {{1, 0, 0}, {0, 1, 0}, {0, 0, 1}};
    double &operator()(int r, int c) { return m[r][c]; }
    double operator()(int r, int c) const { return m[r][c]; }
    // Eigen::Matrix3d::eulerAngles(0,1,2): R = Rx(a) * Ry(b) * Rz(c), a in [0, pi]
    Vector3d eulerAngles(int a0, int a1, int a2) const;
};

struct Quaterniond {
    double qw = 1, qx = 0, qy = 0, qz = 0;
    Quaterniond() = default;
    Quaterniond(double w, double x, double y, double z) : qw(w), qx(x), qy(y), qz(z) {}
    explicit Quaterniond(const Matrix3d &R);
    double w() const { return qw; }
    double x() const { return qx; }
    double y() const { return qy; }
    double z() const { return qz; }
    void normalize()
    {
        const double n = std::sqrt(qw * qw + qx * qx + qy * qy + qz * qz);
        qw /= n, qx /= n, qy /= n, qz /= n;
    }
    Matrix3d toRotationMatrix() const;
    // Hamilton product: (*this * o) rotates by o first, then by *this
    Quaterniond operator*(const Quaterniond &o) const
    {
        return Quaterniond(qw * o.qw - qx * o.qx - qy * o.qy - qz * o.qz, qw * o.qx + qx * o.qw + qy * o.qz - qz * o.qy,
                           qw * o.qy - qx * o.qz + qy * o.qw + qz * o.qx, qw * o.qz + qx * o.qy - qy * o.qx + qz * o.qw);
    }
};

// rotation by `angle` about a UNIT axis; products of rotations are quaternions, like Eigen's AngleAxisd * AngleAxisd
struct AngleAxisd {
    double angle = 0;
    Vector3d axis = Vector3d(1, 0, 0);
    AngleAxisd() = default;
    AngleAxisd(double a, const Vector3d &ax) : angle(a), axis(ax) {}
    operator Quaterniond() const
    {
        const double h = 0.5 * angle, s = std::sin(h);
        return Quaterniond(std::cos(h), s * axis.v[0], s * axis.v[1], s * axis.v[2]);
    }
    Quaterniond operator*(const AngleAxisd &o) const { return Quaterniond(*this) * Quaterniond(o); }
    Quaterniond operator*(const Quaterniond &o) const { return Quaterniond(*this) * o; }
};
inline Quaterniond operator*(const Quaterniond &q, const AngleAxisd &a) { return q * Quaterniond(a); }

// rigid transform [R|t]; operator* composes (this applied after rhs), like Eigen::Affine3d.  Members as in Eigen:
// linear() / translation() (writable), rotation() (by value), Identity(), operator*
class Affine3d {
public:
    static Affine3d Identity() { return Affine3d(); }
    Matrix3d &linear() { return R_; }
    const Matrix3d &linear() const { return R_; }
    Matrix3d rotation() const { return R_; }
    Vector3d &translation() { return t_; }
    const Vector3d &translation() const { return t_; }
    Affine3d operator*(const Affine3d &o) const
    {
        Affine3d r;
        for (int a = 0; a < 3; a++) {
            for (int b = 0; b < 3; b++) {
                double acc = 0;
                for (int k = 0; k < 3; k++) acc += R_.m[a][k] * o.R_.m[k][b];
                r.R_.m[a][b] = acc;
            }
            r.t_.v[a] = R_.m[a][0] * o.t_.v[0] + R_.m[a][1] * o.t_.v[1] + R_.m[a][2] * o.t_.v[2] + t_.v[a];
        }
        return r;
    }

private:
    Matrix3d R_;
    Vector3d t_;
};

template <class T>
struct Triplet {
    int r = 0, c = 0;
    T v = T();
    Triplet() = default;
    Triplet(int row, int col, T val = T()) : r(row), c(col), v(val) {}
    int row() const { return r; }
    int col() const { return c; }
    T value() const { return v; }
};

struct RowMajorTag {};
constexpr int RowMajor = 1;

// compressed row-major sparse matrix with the members the reference uses: rows/cols/outerSize,
// setFromTriplets (ascending columns per row, explicit zeros kept), InnerIterator, coeff
template <class T, int Options = RowMajor>
class SparseMatrix {
public:
    SparseMatrix() = default;
    SparseMatrix(long rows, long cols) : rows_(rows), cols_(cols), outer_(static_cast<std::size_t>(rows) + 1, 0) {}
    long rows() const { return rows_; }
    long cols() const { return cols_; }
    long outerSize() const { return rows_; }
    long nonZeros() const { return static_cast<long>(inner_.size()); }
    template <class It>
    void setFromTriplets(It first, It last);
    void makeCompressed() {}
    bool isCompressed() const { return true; }  // (this stand-in only has the compressed form)
    T coeff(long r, long c) const
    {
        for (int k = outer_[r]; k < outer_[r + 1]; k++)
            if (inner_[k] == c) return values_[k];
        return T();
    }
    const int *outerIndexPtr() const { return outer_.data(); }
    const int *innerIndexPtr() const { return inner_.data(); }
    const T *valuePtr() const { return values_.data(); }
    T *valuePtr() { return values_.data(); }
    void resize(long rows, long cols)
    {
        rows_ = rows, cols_ = cols;
        outer_.assign(static_cast<std::size_t>(rows) + 1, 0);
        inner_.clear(), values_.clear();
    }
    class InnerIterator {
    public:
        InnerIterator(const SparseMatrix &m, long outer) : m_(m), row_(outer), k_(m.outer_[outer]), end_(m.outer_[outer + 1]) {}
        explicit operator bool() const { return k_ < end_; }
        InnerIterator &operator++()
        {
            ++k_;
            return *this;
        }
        long row() const { return row_; }
        long col() const { return m_.inner_[k_]; }
        T value() const { return m_.values_[k_]; }
        long index() const { return k_; }

    private:
        const SparseMatrix &m_;
        long row_;
        int k_, end_;
    };

private:
    long rows_ = 0, cols_ = 0;
    std::vector<int> outer_{0};
    std::vector<int> inner_;
    std::vector<T> values_;
};

}  // namespace Eigen

namespace ceres {
// rho(s) and its derivatives; ScaledLoss(NULL, a): a * s; LossFunctionWrapper: a loss that can be swapped while a
// problem holds it — the three classes ErrorTerm::weight() / updateWeight() are written against (error_term.hpp:17-19,39-45)
enum Ownership { DO_NOT_TAKE_OWNERSHIP, TAKE_OWNERSHIP };
class LossFunction {
public:
    virtual ~LossFunction() {}
    virtual void Evaluate(double sq_norm, double out[3]) const = 0;
};
class ScaledLoss : public LossFunction {
public:
    ScaledLoss(const LossFunction *rho, double a, Ownership ownership) : rho_(rho), a_(a), ownership_(ownership) {}
    ~ScaledLoss() override
    {
        if (ownership_ == TAKE_OWNERSHIP) delete rho_;
    }
    void Evaluate(double s, double out[3]) const override
    {
        if (rho_ == nullptr) {
            out[0] = a_ * s, out[1] = a_, out[2] = 0.0;
        } else {
            rho_->Evaluate(s, out);
            out[0] *= a_, out[1] *= a_, out[2] *= a_;
        }
    }

private:
    const LossFunction *rho_;
    double a_;
    Ownership ownership_;
};
class LossFunctionWrapper : public LossFunction {
public:
    LossFunctionWrapper(LossFunction *rho, Ownership ownership) : rho_(rho), ownership_(ownership) {}
    ~LossFunctionWrapper() override
    {
        if (ownership_ == TAKE_OWNERSHIP) delete rho_;
    }
    LossFunctionWrapper(const LossFunctionWrapper &) = delete;
    LossFunctionWrapper &operator=(const LossFunctionWrapper &) = delete;
    void Evaluate(double sq_norm, double out[3]) const override
    {
        if (rho_ == nullptr) out[0] = sq_norm, out[1] = 1.0, out[2] = 0.0;
        else rho_->Evaluate(sq_norm, out);
    }
    void Reset(LossFunction *rho, Ownership ownership)
    {
        if (ownership_ == TAKE_OWNERSHIP) delete rho_;
        rho_ = rho, ownership_ = ownership;
    }

private:
    LossFunction *rho_;
    Ownership ownership_;
};
enum LinearSolverType { DENSE_QR, SPARSE_NORMAL_CHOLESKY };
// the callback protocol WeightUpdaterCallback is written against (weight_updater_callback.hpp:15,36)
enum CallbackReturnType { SOLVER_CONTINUE, SOLVER_ABORT, SOLVER_TERMINATE_SUCCESSFULLY };
struct IterationSummary {
    int iteration = 0;
    bool step_is_successful = true;
    double cost = 0, cost_change = 0;
};
class IterationCallback {
public:
    virtual ~IterationCallback() {}
    virtual CallbackReturnType operator()(const IterationSummary &summary) = 0;
};
struct Solver {
    struct Options {
        LinearSolverType linear_solver_type = DENSE_QR;  // accepted, unused: the solve is closed form
        bool use_nonmonotonic_steps = false;              // accepted, unused
        bool minimizer_progress_to_stdout = false;
        int max_num_iterations = 50;
        double function_tolerance = 1e-6;
        int num_threads = 1;                              // accepted, unused: the work runs on the GPU
    };
    struct Summary {
        double initial_cost = 0, final_cost = 0;
        int num_successful_steps = 0;
        std::string FullReport() const;
    };
};
}  // namespace ceres

#include "compat_impl.hpp"

#endif  // PPCR_NO_COMPAT_TYPES
