#include "prob_point_cloud_registration/pcd_io.hpp"

#include <algorithm>
#include <cstdint>
#include <exception>
#include <cstring>
#include <fstream>
#include <iomanip>
#include <iostream>
#include <sstream>
#include <vector>

namespace prob_point_cloud_registration {
namespace io {

namespace {
struct Field {
    std::string name;
    int size = 4;
    char type = 'F';
    int count = 1;
    long long offset = 0;  // bytes from the start of a point's record
};

// LZF decoder (the stream format of liblzf, which PCL bundles for DATA binary_compressed).  A control byte c:
//   c < 32        -> copy the next c + 1 bytes literally;
//   otherwise     -> back reference: length = c >> 5 (7 means "add the next byte"), then + 2;
//                    distance = ((c & 31) << 8 | next byte) + 1 behind the write position (may overlap it).
// Returns the number of bytes produced, 0 on a malformed or over-long stream.
std::size_t lzfDecompress(const unsigned char *in, std::size_t in_len, unsigned char *out, std::size_t out_cap)
{
    std::size_t ip = 0, op = 0;
    while (ip < in_len) {
        const unsigned ctrl = in[ip++];
        if (ctrl < 32) {
            const std::size_t run = ctrl + 1;
            if (ip + run > in_len || op + run > out_cap) return 0;
            std::memcpy(out + op, in + ip, run);
            ip += run;
            op += run;
        } else {
            std::size_t len = ctrl >> 5;
            if (len == 7) {
                if (ip >= in_len) return 0;
                len += in[ip++];
            }
            if (ip >= in_len) return 0;
            const std::size_t dist = (static_cast<std::size_t>(ctrl & 31u) << 8 | in[ip++]) + 1;
            len += 2;
            if (dist > op || op + len > out_cap) return 0;
            for (std::size_t k = 0; k < len; k++, op++) out[op] = out[op - dist];  // byte-wise: source may overlap
        }
    }
    return op;
}

std::vector<std::string> split(const std::string &line)
{
    std::istringstream is(line);
    std::vector<std::string> out;
    std::string tok;
    while (is >> tok) out.push_back(tok);
    return out;
}

// bytes left in the stream from the current read position (-1 when the stream cannot seek)
long long remainingBytes(std::ifstream &in)
{
    const std::streampos here = in.tellg();
    if (here < 0) return -1;
    in.seekg(0, std::ios::end);
    const std::streampos end = in.tellg();
    in.seekg(here);
    return (end < 0 || !in) ? -1 : static_cast<long long>(end - here);
}

int loadChecked(const std::string &file_name, pcl::PointCloud<pcl::PointXYZ> &cloud)
{
    std::ifstream in(file_name, std::ios::binary);
    if (!in) {
        std::cerr << "[pcd] cannot open " << file_name << std::endl;
        return -1;
    }
    std::vector<Field> fields;
    long long width = 0, height = 1, points = -1;
    std::string data_mode, line;
    while (std::getline(in, line)) {
        if (!line.empty() && line.back() == '\r') line.pop_back();
        if (line.empty() || line[0] == '#') continue;
        const std::vector<std::string> tok = split(line);
        if (tok.empty()) continue;
        const std::string &key = tok[0];
        if (key == "FIELDS" || key == "COLUMNS") {
            fields.resize(tok.size() - 1);
            for (std::size_t i = 1; i < tok.size(); i++) fields[i - 1].name = tok[i];
        } else if (key == "SIZE") {
            for (std::size_t i = 1; i < tok.size() && i - 1 < fields.size(); i++) fields[i - 1].size = std::stoi(tok[i]);
        } else if (key == "TYPE") {
            for (std::size_t i = 1; i < tok.size() && i - 1 < fields.size(); i++) fields[i - 1].type = tok[i][0];
        } else if (key == "COUNT") {
            for (std::size_t i = 1; i < tok.size() && i - 1 < fields.size(); i++) fields[i - 1].count = std::stoi(tok[i]);
        } else if (key == "WIDTH") {
            width = std::stoll(tok.at(1));
        } else if (key == "HEIGHT") {
            height = std::stoll(tok.at(1));
        } else if (key == "POINTS") {
            points = std::stoll(tok.at(1));
        } else if (key == "DATA") {
            data_mode = tok.size() > 1 ? tok[1] : "";
            break;
        }
    }
    if (points < 0 && width >= 0 && height >= 0 && (height == 0 || width <= (1ll << 40) / std::max(height, 1ll))) points = width * height;
    for (const Field &f : fields)
        if (f.size <= 0 || f.size > 8 || f.count <= 0 || f.count > (1 << 20)) {
            std::cerr << "[pcd] " << file_name << ": bad SIZE / COUNT in the header" << std::endl;
            return -1;
        }
    // (64-bit running sums: 400 fields of 8 x 2^20 bytes wrap an int — found by the UBSan run of tests/test_sanitizers.py)
    int ix = -1, iy = -1, iz = -1;
    long long col = 0, off = 0;
    std::vector<long long> first_col(fields.size(), 0);
    for (std::size_t f = 0; f < fields.size(); f++) {
        fields[f].offset = off;
        first_col[f] = col;
        off += static_cast<long long>(fields[f].size) * fields[f].count;
        col += fields[f].count;
        if (fields[f].name == "x") ix = static_cast<int>(f);
        if (fields[f].name == "y") iy = static_cast<int>(f);
        if (fields[f].name == "z") iz = static_cast<int>(f);
    }
    if (ix < 0 || iy < 0 || iz < 0 || points < 0) {
        std::cerr << "[pcd] " << file_name << ": header lacks x/y/z fields or a point count" << std::endl;
        return -1;
    }
    if (off > (1ll << 24) || col > (1ll << 22)) {
        std::cerr << "[pcd] " << file_name << ": a point record of " << off << " bytes / " << col << " columns is refused" << std::endl;
        return -1;
    }
    for (int f : {ix, iy, iz})
        if (fields[f].type != 'F' || fields[f].size != 4) {
            std::cerr << "[pcd] " << file_name << ": x/y/z must be float32" << std::endl;
            return -1;
        }
    // a point needs at least one byte of data whatever the encoding's best case (LZF expands <= ~ 264x): a header
    // that promises more points than the file could hold is refused before anything is allocated
    const long long left = remainingBytes(in);
    if (left >= 0 && (points > left * 512 + 16 || (data_mode == "binary" && points > left / std::max(off, 1ll)))) {
        std::cerr << "[pcd] " << file_name << ": header promises more points than the file holds" << std::endl;
        return -1;
    }
    cloud.points.assign(static_cast<std::size_t>(points), pcl::PointXYZ());
    if (data_mode == "ascii") {
        const long long ncols = col;
        std::vector<double> row(static_cast<std::size_t>(ncols));
        for (long long i = 0; i < points; i++) {
            if (!std::getline(in, line)) {
                std::cerr << "[pcd] " << file_name << ": truncated ascii data" << std::endl;
                return -1;
            }
            std::istringstream is(line);
            for (long long k = 0; k < ncols; k++) {
                std::string tok;
                if (!(is >> tok)) {
                    std::cerr << "[pcd] " << file_name << ": short ascii row " << i << std::endl;
                    return -1;
                }
                row[static_cast<std::size_t>(k)] = (tok == "nan" || tok == "NaN") ? NAN : std::strtod(tok.c_str(), nullptr);
            }
            pcl::PointXYZ &p = cloud.points[static_cast<std::size_t>(i)];
            p.x = static_cast<float>(row[static_cast<std::size_t>(first_col[ix])]);
            p.y = static_cast<float>(row[static_cast<std::size_t>(first_col[iy])]);
            p.z = static_cast<float>(row[static_cast<std::size_t>(first_col[iz])]);
        }
    } else if (data_mode == "binary") {
        const std::size_t stride = static_cast<std::size_t>(off);
        std::vector<char> buf(stride * static_cast<std::size_t>(points));
        in.read(buf.data(), static_cast<std::streamsize>(buf.size()));
        if (static_cast<std::size_t>(in.gcount()) != buf.size()) {
            std::cerr << "[pcd] " << file_name << ": truncated binary data" << std::endl;
            return -1;
        }
        for (long long i = 0; i < points; i++) {
            const char *rec = buf.data() + static_cast<std::size_t>(i) * stride;
            pcl::PointXYZ &p = cloud.points[static_cast<std::size_t>(i)];
            std::memcpy(&p.x, rec + fields[ix].offset, 4);
            std::memcpy(&p.y, rec + fields[iy].offset, 4);
            std::memcpy(&p.z, rec + fields[iz].offset, 4);
        }
    } else if (data_mode == "binary_compressed") {
        // uint32 compressed size, uint32 uncompressed size, LZF stream; the payload is a struct of arrays: all
        // values of field 0 for every point, then field 1, ... (each field block is size * count * points bytes)
        std::uint32_t sizes[2] = {0, 0};
        in.read(reinterpret_cast<char *>(sizes), 8);
        if (in.gcount() != 8) {
            std::cerr << "[pcd] " << file_name << ": truncated binary_compressed header" << std::endl;
            return -1;
        }
        const std::size_t expect = static_cast<std::size_t>(off) * static_cast<std::size_t>(points);
        if (sizes[1] != expect) {
            std::cerr << "[pcd] " << file_name << ": uncompressed size " << sizes[1] << " does not match the header (" << expect
                      << ")" << std::endl;
            return -1;
        }
        if (left >= 0 && static_cast<long long>(sizes[0]) > left - 8) {
            std::cerr << "[pcd] " << file_name << ": truncated binary_compressed data" << std::endl;
            return -1;
        }
        std::vector<unsigned char> comp(sizes[0]), raw(expect);
        in.read(reinterpret_cast<char *>(comp.data()), static_cast<std::streamsize>(comp.size()));
        if (static_cast<std::size_t>(in.gcount()) != comp.size()) {
            std::cerr << "[pcd] " << file_name << ": truncated binary_compressed data" << std::endl;
            return -1;
        }
        if (expect > 0 && lzfDecompress(comp.data(), comp.size(), raw.data(), raw.size()) != expect) {
            std::cerr << "[pcd] " << file_name << ": corrupt LZF stream" << std::endl;
            return -1;
        }
        const std::size_t np = static_cast<std::size_t>(points);
        auto block = [&](int f) { return raw.data() + static_cast<std::size_t>(fields[f].offset) * np; };
        const std::size_t sx = static_cast<std::size_t>(fields[ix].size * fields[ix].count);
        const std::size_t sy = static_cast<std::size_t>(fields[iy].size * fields[iy].count);
        const std::size_t sz = static_cast<std::size_t>(fields[iz].size * fields[iz].count);
        for (std::size_t i = 0; i < np; i++) {
            pcl::PointXYZ &p = cloud.points[i];
            std::memcpy(&p.x, block(ix) + i * sx, 4);
            std::memcpy(&p.y, block(iy) + i * sy, 4);
            std::memcpy(&p.z, block(iz) + i * sz, 4);
        }
    } else {
        std::cerr << "[pcd] " << file_name << ": DATA " << data_mode << " is not supported (ascii, binary and binary_compressed are)"
                  << std::endl;
        return -1;
    }
    return 0;
}

}  // namespace

int loadPCDFile(const std::string &file_name, pcl::PointCloud<pcl::PointXYZ> &cloud)
{
    // pcl::io::loadPCDFile returns -1 on anything it cannot read; nothing may escape to the CLI
    try {
        return loadChecked(file_name, cloud);
    } catch (const std::exception &e) {
        std::cerr << "[pcd] " << file_name << ": malformed file (" << e.what() << ")" << std::endl;
        cloud.points.clear();
        return -1;
    }
}

int savePCDFile(const std::string &file_name, const pcl::PointCloud<pcl::PointXYZ> &cloud, bool binary_mode)
{
    std::ofstream out(file_name, std::ios::binary);
    if (!out) return -1;
    out << "# .PCD v0.7 - Point Cloud Data file format\nVERSION 0.7\nFIELDS x y z\nSIZE 4 4 4\nTYPE F F F\nCOUNT 1 1 1\n"
        << "WIDTH " << cloud.size() << "\nHEIGHT 1\nVIEWPOINT 0 0 0 1 0 0 0\nPOINTS " << cloud.size() << "\nDATA "
        << (binary_mode ? "binary" : "ascii") << "\n";
    if (binary_mode) {
        for (const auto &p : cloud.points) out.write(reinterpret_cast<const char *>(&p.x), 12);
    } else {
        out << std::setprecision(9);
        for (const auto &p : cloud.points) out << p.x << " " << p.y << " " << p.z << "\n";
    }
    return out ? 0 : -1;
}

}  // namespace io
}  // namespace prob_point_cloud_registration
