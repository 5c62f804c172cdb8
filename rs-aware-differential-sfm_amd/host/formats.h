// formats.h -- on-disk formats of the reference's example archives and outputs, host-side only:
//   CSV matrices (A.csv, N_rs_t.csv, N_rs_r.csv, N_rs_unproject_{x,y,z}.csv)   camera.cc:99-176, rsframe.cc:58-218, :444-553
//   ascii PLY point cloud                                                      camera.cc:423-491
//   8-bit PNG frames (non-interlaced grey / RGB[A]; needs -lz)                 cv::imread / cv::imwrite in main.cc:613-671, :526-556
// Numbers are parsed with ::atof like the reference; a file whose number of '\n' differs from the expected number of lines
// is rejected (the reference prints a message and leaves the object unset; here the loaders return false).
// Included by camera.h / rsframe.h users who want the file-based members; PNG support is compiled only when
// RSDSFM_WITH_PNG is defined (it pulls in zlib).
#ifndef RSDSFM_HOST_FORMATS_H
#define RSDSFM_HOST_FORMATS_H

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iomanip>
#include <sstream>
#include <string>
#include <vector>

#include "camera.h"

#ifdef RSDSFM_WITH_PNG
#include <zlib.h>
#endif

namespace rsdsfm {

// rows x cols numbers, row-major; false if the file is missing or its line count is not `rows`
inline bool read_matrix_csv(const std::string& path, long rows, long cols, std::vector<double>& out) {
    std::ifstream in(path, std::ios::binary);
    if (!in) return false;
    std::stringstream ss;
    ss << in.rdbuf();
    const std::string txt = ss.str();
    long nl = 0;
    for (char ch : txt) nl += ch == '\n';
    if (nl != rows) return false;
    out.assign((size_t)(rows * cols), 0.0);
    size_t pos = 0;
    for (long i = 0; i < rows; ++i) {
        const size_t eol = txt.find('\n', pos);
        const std::string line = txt.substr(pos, eol - pos);
        pos = eol + 1;
        size_t lp = 0;
        for (long j = 0; j < cols; ++j) {  // cols-1 comma-terminated fields, the rest of the line is the last field
            size_t comma = (j < cols - 1) ? line.find(',', lp) : std::string::npos;
            const std::string field = lp <= line.size() ? line.substr(lp, comma == std::string::npos ? std::string::npos : comma - lp) : std::string();
            out[(size_t)(i * cols + j)] = ::atof(field.c_str());
            if (comma == std::string::npos) {
                lp = line.size() + 1;
            } else {
                lp = comma + 1;
            }
        }
    }
    return true;
}

/** reference camera.cc:99-176 */
inline bool loadIntrinsicsFromFile(Camera& camera, const std::string& csv_intrinsic_matrix) {
    std::vector<double> k;
    if (!read_matrix_csv(csv_intrinsic_matrix, 3, 3, k)) return false;
    lite::Matrix3d K = lite::Matrix3d::Zero();
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) K(r, c) = k[(size_t)(r * 3 + c)];
    camera.setIntrinsics(K);
    return true;
}

/** reference rsframe.cc:444-553: absolute AND relative pose of every scanline from the two CSV files */
inline bool setPoses(RsFrame& frame, const std::string& csv_poses, const std::string& csv_orientation) {
    const long rows = frame.getRows();
    std::vector<double> t, R;
    if (!read_matrix_csv(csv_poses, rows, 3, t) || !read_matrix_csv(csv_orientation, rows, 9, R)) return false;
    for (long i = 0; i < rows; ++i) {
        lite::Matrix3d Ri;
        for (int r = 0; r < 3; ++r)
            for (int c = 0; c < 3; ++c) Ri(r, c) = R[(size_t)(i * 9 + r * 3 + c)];
        const lite::Vector3d ti(t[(size_t)(i * 3)], t[(size_t)(i * 3 + 1)], t[(size_t)(i * 3 + 2)]);
        Scanline& sl = frame.scanline((int)i);
        sl.setRotation(Ri), sl.setRelativeRotation(Ri);
        sl.setTranslation(ti), sl.setRelativeTranslation(ti);
    }
    return true;
}

/** reference rsframe.cc:58-218 */
inline bool setUnprojectionMapRs(RsFrame& frame, const std::string& csv_x, const std::string& csv_y, const std::string& csv_z) {
    const long rows = frame.getRows(), cols = frame.getCols();
    lite::MatrixXd m[3] = {lite::MatrixXd(rows, cols), lite::MatrixXd(rows, cols), lite::MatrixXd(rows, cols)};
    const std::string* paths[3] = {&csv_x, &csv_y, &csv_z};
    for (int a = 0; a < 3; ++a) {
        std::vector<double> v;
        if (!read_matrix_csv(*paths[a], rows, cols, v)) return false;
        for (long i = 0; i < rows; ++i)
            for (long j = 0; j < cols; ++j) m[a](i, j) = v[(size_t)(i * cols + j)];
    }
    return frame.setUnprojectionMapRs(m[0], m[1], m[2]);
}

/** reference camera.cc:423-491: ascii PLY of the frame's 3-D coordinates coloured by the RS image (BGR -> RGB) */
inline bool createPointCloud(Camera& camera, const int frameNr, const std::string& fileName) {
    RsFrame frame = camera.getFrame(frameNr);
    const ImageXYZf coords = frame.get3dCoordinates();
    const ImageBGR colors = frame.getRsImage();
    const size_t total = (size_t)coords.rows() * (size_t)coords.cols();
    if (total != (size_t)colors.rows() * (size_t)colors.cols()) return false;
    std::ofstream out(fileName);
    if (!out) return false;
    out << "ply\nformat ascii 1.0\ncomment PLY File created by RS aware SfM wrapper\nelement vertex " << total
        << "\nproperty float x\nproperty float y\nproperty float z\nproperty uchar red\nproperty uchar green\nproperty uchar blue\nend_header\n";
    const float* p = coords.data();
    const unsigned char* c = colors.data();
    for (size_t i = 0; i < 3 * total; i += 3) {
        for (int j = 0; j < 3; ++j) out << std::setprecision(9) << p[i + (size_t)j] << " ";
        out << (unsigned short)c[i + 2] << " " << (unsigned short)c[i + 1] << " " << (unsigned short)c[i] << "\n";
    }
    return (bool)out;
}

#ifdef RSDSFM_WITH_PNG
namespace png_detail {
inline void put32(std::string& s, unsigned v) {
    for (int sh = 24; sh >= 0; sh -= 8) s.push_back((char)((v >> sh) & 0xff));
}
inline void chunk(std::string& out, const char tag[4], const std::string& data) {
    put32(out, (unsigned)data.size());
    std::string td(tag, 4);
    td += data;
    out += td;
    put32(out, (unsigned)crc32(0L, reinterpret_cast<const Bytef*>(td.data()), (uInt)td.size()));
}
inline unsigned get32(const unsigned char* p) { return ((unsigned)p[0] << 24) | ((unsigned)p[1] << 16) | ((unsigned)p[2] << 8) | p[3]; }
inline bool write(const std::string& path, const unsigned char* px, int rows, int cols, int ch, int level) {
    std::string raw;
    raw.reserve((size_t)rows * ((size_t)cols * (size_t)ch + 1));
    for (int r = 0; r < rows; ++r) {
        raw.push_back(0);  // filter type 0
        for (int c = 0; c < cols; ++c)
            for (int k = 0; k < ch; ++k) raw.push_back((char)px[((size_t)r * (size_t)cols + (size_t)c) * (size_t)ch + (size_t)(ch == 3 ? 2 - k : k)]);  // BGR -> RGB
    }
    uLongf zlen = compressBound((uLong)raw.size());
    std::string z(zlen, '\0');
    if (compress2(reinterpret_cast<Bytef*>(&z[0]), &zlen, reinterpret_cast<const Bytef*>(raw.data()), (uLong)raw.size(), level) != Z_OK) return false;
    z.resize(zlen);
    std::string out("\x89PNG\r\n\x1a\n", 8), ihdr;
    put32(ihdr, (unsigned)cols), put32(ihdr, (unsigned)rows);
    ihdr.push_back(8), ihdr.push_back((char)(ch == 3 ? 2 : 0)), ihdr.push_back(0), ihdr.push_back(0), ihdr.push_back(0);
    chunk(out, "IHDR", ihdr), chunk(out, "IDAT", z), chunk(out, "IEND", std::string());
    std::ofstream f(path, std::ios::binary);
    f.write(out.data(), (std::streamsize)out.size());
    return (bool)f;
}
// decodes into interleaved channels as stored (1, 2, 3 or 4 per pixel)
inline bool read(const std::string& path, std::vector<unsigned char>& px, int& rows, int& cols, int& ch) {
    std::ifstream in(path, std::ios::binary);
    if (!in) return false;
    std::stringstream ss;
    ss << in.rdbuf();
    const std::string d = ss.str();
    if (d.size() < 8 || std::memcmp(d.data(), "\x89PNG\r\n\x1a\n", 8) != 0) return false;
    const unsigned char* u = reinterpret_cast<const unsigned char*>(d.data());
    size_t pos = 8;
    std::string idat;
    int ctype = -1, depth = 0, interlace = 0;
    while (pos + 12 <= d.size()) {
        const unsigned len = get32(u + pos);
        const std::string tag(d.data() + pos + 4, 4);
        if (tag == "IHDR") {
            cols = (int)get32(u + pos + 8), rows = (int)get32(u + pos + 12);
            depth = u[pos + 16], ctype = u[pos + 17], interlace = u[pos + 20];
        } else if (tag == "IDAT") {
            idat.append(d.data() + pos + 8, len);
        } else if (tag == "IEND") {
            break;
        }
        pos += 12 + (size_t)len;
    }
    if (depth != 8 || interlace != 0) return false;
    ch = ctype == 0 ? 1 : ctype == 2 ? 3 : ctype == 4 ? 2 : ctype == 6 ? 4 : 0;
    if (!ch) return false;
    const size_t stride = (size_t)cols * (size_t)ch;
    std::vector<unsigned char> raw((size_t)rows * (stride + 1));
    uLongf rl = (uLongf)raw.size();
    if (uncompress(raw.data(), &rl, reinterpret_cast<const Bytef*>(idat.data()), (uLong)idat.size()) != Z_OK || rl != raw.size()) return false;
    px.assign((size_t)rows * stride, 0);
    for (int r = 0; r < rows; ++r) {
        const unsigned char ft = raw[(size_t)r * (stride + 1)];
        const unsigned char* line = &raw[(size_t)r * (stride + 1) + 1];
        unsigned char* cur = &px[(size_t)r * stride];
        const unsigned char* prev = r ? &px[(size_t)(r - 1) * stride] : nullptr;
        for (size_t i = 0; i < stride; ++i) {
            const int a = i >= (size_t)ch ? cur[i - (size_t)ch] : 0, b = prev ? prev[i] : 0, c = (prev && i >= (size_t)ch) ? prev[i - (size_t)ch] : 0;
            int pred = 0;
            if (ft == 1) pred = a;
            else if (ft == 2) pred = b;
            else if (ft == 3) pred = (a + b) >> 1;
            else if (ft == 4) {
                const int pa = std::abs(b - c), pb = std::abs(a - c), pc = std::abs(a + b - 2 * c);
                pred = (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c);
            } else if (ft != 0) return false;
            cur[i] = (unsigned char)((line[i] + pred) & 255);
        }
    }
    return true;
}
}  // namespace png_detail

/** cv::imwrite(path, mat, {CV_IMWRITE_PNG_COMPRESSION, level}) for an 8-bit BGR image */
inline bool imwrite_png(const std::string& path, const ImageBGR& img, int level = 0) { return png_detail::write(path, img.data(), img.rows(), img.cols(), 3, level); }
inline bool imwrite_png_gray(const std::string& path, const std::vector<unsigned char>& img, int rows, int cols, int level = 0) {
    return (size_t)rows * (size_t)cols == img.size() && png_detail::write(path, img.data(), rows, cols, 1, level);
}
/** cv::imread(path, CV_LOAD_IMAGE_COLOR): grey files are replicated to three channels, alpha is dropped */
inline bool imread_png(const std::string& path, ImageBGR& img) {
    std::vector<unsigned char> px;
    int rows = 0, cols = 0, ch = 0;
    if (!png_detail::read(path, px, rows, cols, ch)) return false;
    img = ImageBGR(rows, cols);
    for (int r = 0; r < rows; ++r)
        for (int c = 0; c < cols; ++c) {
            const unsigned char* p = &px[((size_t)r * (size_t)cols + (size_t)c) * (size_t)ch];
            if (ch <= 2) img.at(r, c, 0) = img.at(r, c, 1) = img.at(r, c, 2) = p[0];
            else img.at(r, c, 0) = p[2], img.at(r, c, 1) = p[1], img.at(r, c, 2) = p[0];
        }
    return true;
}
#endif  // RSDSFM_WITH_PNG

}  // namespace rsdsfm
#endif
