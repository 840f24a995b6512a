// verbose-gated stdout (the reference's OutputStream, output_stream.hpp:7-23)
#pragma once
#include <iostream>

namespace prob_point_cloud_registration {

class OutputStream {
public:
    explicit OutputStream(bool verbose = false) : verbose_(verbose) {}
    template <class T>
    OutputStream &operator<<(const T &v)
    {
        if (verbose_) std::cout << v;
        return *this;
    }

private:
    bool verbose_;
};

}  // namespace prob_point_cloud_registration
