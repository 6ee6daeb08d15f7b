// dataset_reader.cpp -- see dataset_reader.h.
#include "dataset_reader.h"

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <iomanip>
#include <iostream>
#include <stdexcept>

#include "image_io.h"

namespace pvio {

// ---- CSV ------------------------------------------------------------------------------------------------------------
// One record per line, first line is a header.  Lines are split at '\n'; EuRoC files end their lines with "\r\n" and the
// carriage return is not part of the file name (the reference's "%2047[^\r]\r\n"), TUM-VI names run to the '\n'.
// Timestamps are nanoseconds parsed as a double (the reference's "%lf") and scaled by 1e-9.
namespace {

bool read_line(FILE *f, std::string &line) {
    line.clear();
    int c;
    bool any = false;
    while ((c = std::fgetc(f)) != EOF) {
        any = true;
        if (c == '\n') break;
        line.push_back((char)c);
    }
    return any;
}

} // namespace

std::vector<CameraCsvItem> load_camera_csv(const std::string &filename, bool euroc) {
    std::vector<CameraCsvItem> items;
    FILE *csv = std::fopen(filename.c_str(), "r");
    if (!csv) return items; // like the reference: a missing file is an empty sequence
    std::string line;
    read_line(csv, line); // header
    while (read_line(csv, line)) {
        if (euroc) {
            const size_t cr = line.find('\r');
            if (cr != std::string::npos) line.resize(cr);
        }
        char *end = nullptr;
        const double t = std::strtod(line.c_str(), &end);
        if (end == line.c_str() || *end != ',' || end[1] == '\0') break; // the reference stops at the first malformed record
        CameraCsvItem item;
        item.t = t * 1e-9;
        item.filename = std::string(end + 1).substr(0, 2047);
        items.push_back(std::move(item));
    }
    std::fclose(csv);
    return items;
}

std::vector<ImuCsvItem> load_imu_csv(const std::string &filename, bool /*euroc*/) {
    std::vector<ImuCsvItem> items;
    FILE *csv = std::fopen(filename.c_str(), "r");
    if (!csv) return items;
    std::string line;
    read_line(csv, line); // header
    while (read_line(csv, line)) {
        double v[7];
        const char *p = line.c_str();
        int k = 0;
        for (; k < 7; ++k) {
            char *end = nullptr;
            v[k] = std::strtod(p, &end);
            if (end == p) break;
            p = end;
            if (k < 6) {
                if (*p != ',') break;
                ++p;
            }
        }
        if (k != 7) break;
        ImuCsvItem item;
        item.t = v[0] * 1e-9;
        for (int i = 0; i < 3; ++i) item.w[i] = v[1 + i], item.a[i] = v[4 + i];
        items.push_back(item);
    }
    std::fclose(csv);
    return items;
}

// ---- images ---------------------------------------------------------------------------------------------------------
UndistortedHipImage::UndistortedHipImage(pvio_hip_ctx *ctx, std::shared_ptr<pvio_hip_undistort> ud, int out_width, int out_height, const uint8_t *pixels,
                                         int width, int height, int stride, double timestamp)
    : HipImage(ctx, pixels, width, height, stride, timestamp), ud_(std::move(ud)), ow_(out_width), oh_(out_height) {}

void UndistortedHipImage::preprocess() {
    if (img_) pvio_hip_image_release(ctx_, img_), img_ = nullptr;
    forget_host_levels();
    const int32_t rc = pvio_hip_image_create_undistorted(ctx_, ud_.get(), pixels_.data(), w_, h_, w_, /*apply_clahe=*/1, &img_);
    if (rc != 0) throw std::runtime_error(std::string("pvio_hip_image_create_undistorted: ") + pvio_hip_last_error(ctx_)); // no CPU path
}

// ---- sequence -------------------------------------------------------------------------------------------------------
SequenceReader::SequenceReader(const std::string &path, bool euroc, pvio_hip_ctx *ctx) : ctx_(ctx) {
    for (auto &item : load_camera_csv(path + "/cam0/data.csv", euroc)) {
        image_data.emplace_back(item.t, path + "/cam0/data/" + item.filename);
        all_data.emplace_back(item.t, NextDataType::CAMERA);
    }
    for (auto &item : load_imu_csv(path + "/imu0/data.csv", euroc)) {
        vector<3> gyr, acc;
        for (int i = 0; i < 3; ++i) gyr[i] = item.w[i], acc[i] = item.a[i];
        gyroscope_data.emplace_back(item.t, gyr);
        all_data.emplace_back(item.t, NextDataType::GYROSCOPE);
        accelerometer_data.emplace_back(item.t, acc);
        all_data.emplace_back(item.t, NextDataType::ACCELEROMETER);
    }
    // the reference sorts with std::sort on the timestamp only; events with EQUAL timestamps (every IMU row yields a
    // gyroscope and an accelerometer event) then come out in an unspecified order -- a stable sort keeps the insertion
    // order (camera, gyroscope, accelerometer), which is one of the orders std::sort may produce
    auto by_time = [](const auto &a, const auto &b) { return a.first < b.first; };
    std::stable_sort(all_data.begin(), all_data.end(), by_time);
    std::stable_sort(image_data.begin(), image_data.end(), by_time);
    std::stable_sort(gyroscope_data.begin(), gyroscope_data.end(), by_time);
    std::stable_sort(accelerometer_data.begin(), accelerometer_data.end(), by_time);
}

DatasetReader::NextDataType SequenceReader::next() {
    if (all_data.empty()) return NextDataType::END;
    return all_data.front().second;
}

std::shared_ptr<Image> SequenceReader::read_image() {
    if (image_data.empty()) return nullptr;
    const auto [t, filename] = image_data.front();
    const GrayImage img = read_gray_image(filename);
    if (!ud_ || ud_w_ != img.width || ud_h_ != img.height) {
        const FixedRemap &m = maps_for(img.width, img.height);
        pvio_hip_undistort *raw = nullptr;
        if (pvio_hip_undistort_create(ctx_, m.xy.data(), m.frac.data(), m.width, m.height, &raw) != 0)
            throw std::runtime_error(std::string("pvio_hip_undistort_create: ") + pvio_hip_last_error(ctx_));
        pvio_hip_ctx *ctx = ctx_;
        ud_ = std::shared_ptr<pvio_hip_undistort>(raw, [ctx](pvio_hip_undistort *u) { pvio_hip_undistort_release(ctx, u); });
        ud_w_ = img.width, ud_h_ = img.height;
    }
    auto out = std::make_shared<UndistortedHipImage>(ctx_, ud_, ud_w_, ud_h_, img.pixels.data(), img.width, img.height, img.width, t);
    all_data.pop_front();
    image_data.pop_front();
    return out;
}

std::pair<double, vector<3>> SequenceReader::read_gyroscope() {
    if (gyroscope_data.empty()) return {};
    auto item = gyroscope_data.front();
    all_data.pop_front();
    gyroscope_data.pop_front();
    return item;
}

std::pair<double, vector<3>> SequenceReader::read_accelerometer() {
    if (accelerometer_data.empty()) return {};
    auto item = accelerometer_data.front();
    all_data.pop_front();
    accelerometer_data.pop_front();
    return item;
}

const FixedRemap &EurocDatasetReader::maps_for(int width, int height) {
    // euroc_dataset_reader.cpp:73-74: float32 matrices handed to cv::undistort
    static const float dist[4] = {-0.28340811f, 0.07395907f, 0.00019359f, 1.76187114e-05f};
    static const float K[9] = {458.654f, 0, 367.215f, 0, 457.296f, 248.375f, 0, 0, 1};
    maps_ = cv_undistort_fixed_maps(K, dist, 4, width, height);
    return maps_;
}

const FixedRemap &TUMDatasetReader::maps_for(int width, int height) {
    // tum_dataset_reader.cpp:74-79
    matrix<3> K;
    K(0, 0) = 190.97847715128717, K(0, 2) = 254.93170605935475;
    K(1, 1) = 190.9733070521226, K(1, 2) = 256.8974428996504;
    K(2, 2) = 1;
    const std::vector<double> dist = {0.0034003170790442797, 0.001766278153469831, -0.00266312569781606, 0.0003299517423931039};
    image_undistorter = std::make_unique<ImageUndistorter>((size_t)width, (size_t)height, K, dist, "equidistant");
    return image_undistorter->maps();
}

namespace {
bool strip_scheme(const std::string &s, const std::string &scheme, std::string &rest) {
    if (s.compare(0, scheme.size(), scheme) != 0) return false;
    rest = s.substr(scheme.size());
    return true;
}
} // namespace

std::unique_ptr<DatasetReader> DatasetReader::create_reader(const std::string &filename, pvio_hip_ctx *ctx) {
    std::string path;
    if (strip_scheme(filename, "euroc://", path)) return std::make_unique<EurocDatasetReader>(path, ctx);
    if (strip_scheme(filename, "tum://", path)) return std::make_unique<TUMDatasetReader>(path, ctx);
    return nullptr; // "sensors://", "legacy-sensors://": not provided
}

// ---- trajectory output ----------------------------------------------------------------------------------------------
TumOutputWriter::TumOutputWriter(const std::string &filename) {
    file.open(filename.c_str());
    if (!file.is_open()) std::cerr << "Cannot open file " << std::quoted(filename) << std::endl;
    file.precision(15);
}

void TumOutputWriter::write_pose(const double &t, const OutputPose &pose) {
    const quaternion &q = pose.q;
    file << t << " " << pose.p[0] << " " << pose.p[1] << " " << pose.p[2] << " " << q.x() << " " << q.y() << " " << q.z() << " " << q.w() << "\n";
    file.flush();
}

} // namespace pvio
