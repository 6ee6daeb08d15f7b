// dataset_reader.h -- the data formats in front of the hot path: EuRoC / TUM-VI sequence
// readers and the TUM trajectory writer of pvio-pc, without OpenCV.  Same interface and event order as
//   DatasetReader            pvio-pc/src/dataset_reader.h:23-38, create_reader dataset_reader.cpp:34-46 ("euroc://", "tum://")
//   EurocDatasetReader       pvio-pc/src/euroc_dataset_reader.{h,cpp}   (CSV formats .h:52,57,102; timestamps ns -> s)
//   TUMDatasetReader         pvio-pc/src/tum_dataset_reader.{h,cpp}     (same layout, '\n' line ends)
//   TumOutputWriter          pvio-pc/src/output_writer.h:32-50          (precision 15, "t px py pz qx qy qz qw")
// read_image() decodes the file on the host, uploads the DISTORTED pixels and undistorts them on the GPU in front of the
// pyramid (pvio_hip_image_create_undistorted) -- the camera constants are the ones hard-coded in the reference readers
// (euroc_dataset_reader.cpp:73-74, tum_dataset_reader.cpp:75-79).  The "sensors://" readers (a proprietary capture format)
// are not provided.
#pragma once
#include <deque>
#include <fstream>
#include <memory>
#include <string>
#include <utility>
#include <vector>

#include "../../include/pvio_hip.h"
#include "feature_front.h"
#include "host_seam.h"
#include "undistort_maps.h"

namespace pvio {

// A HipImage whose pixels are undistorted on the device when the pyramid is built (preprocess()).
class UndistortedHipImage : public HipImage {
  public:
    UndistortedHipImage(pvio_hip_ctx *ctx, std::shared_ptr<pvio_hip_undistort> ud, int out_width, int out_height, const uint8_t *pixels, int width,
                        int height, int stride, double timestamp);
    size_t width() const override { return (size_t)ow_; }
    size_t height() const override { return (size_t)oh_; }
    void preprocess() override;

  private:
    std::shared_ptr<pvio_hip_undistort> ud_;
    int ow_, oh_;
};

class DatasetReader {
  public:
    enum NextDataType { AGAIN, CAMERA, GYROSCOPE, ACCELEROMETER, END };
    virtual ~DatasetReader() = default;
    virtual NextDataType next() = 0;
    virtual std::shared_ptr<Image> read_image() = 0;
    virtual std::pair<double, vector<3>> read_gyroscope() = 0;
    virtual std::pair<double, vector<3>> read_accelerometer() = 0;
    // "euroc://<dir>" or "tum://<dir>" (<dir> holds cam0/data.csv, cam0/data/, imu0/data.csv); nullptr for other schemes
    static std::unique_ptr<DatasetReader> create_reader(const std::string &filename, pvio_hip_ctx *ctx);
};

struct CameraCsvItem {
    double t;
    std::string filename;
};
struct ImuCsvItem {
    double t;
    double w[3], a[3];
};
// euroc = true: "\r\n" line ends (euroc_dataset_reader.h:52-57,100-102); false: "\n" (tum_dataset_reader.h:54-59,102-104)
std::vector<CameraCsvItem> load_camera_csv(const std::string &filename, bool euroc);
std::vector<ImuCsvItem> load_imu_csv(const std::string &filename, bool euroc);

class SequenceReader : public DatasetReader { // the part EurocDatasetReader and TUMDatasetReader share
  public:
    NextDataType next() override;
    std::shared_ptr<Image> read_image() override;
    std::pair<double, vector<3>> read_gyroscope() override;
    std::pair<double, vector<3>> read_accelerometer() override;
    size_t images_left() const { return image_data.size(); }

  protected:
    SequenceReader(const std::string &path, bool euroc, pvio_hip_ctx *ctx);
    virtual const FixedRemap &maps_for(int width, int height) = 0; // built on the first image
    pvio_hip_ctx *ctx_;
    std::shared_ptr<pvio_hip_undistort> ud_;
    int ud_w_ = 0, ud_h_ = 0;
    std::deque<std::pair<double, NextDataType>> all_data;
    std::deque<std::pair<double, vector<3>>> gyroscope_data, accelerometer_data;
    std::deque<std::pair<double, std::string>> image_data;
};

class EurocDatasetReader : public SequenceReader {
  public:
    EurocDatasetReader(const std::string &euroc_path, pvio_hip_ctx *ctx) : SequenceReader(euroc_path, true, ctx) {}

  protected:
    const FixedRemap &maps_for(int width, int height) override;
    FixedRemap maps_;
};

class TUMDatasetReader : public SequenceReader {
  public:
    TUMDatasetReader(const std::string &tum_path, pvio_hip_ctx *ctx) : SequenceReader(tum_path, false, ctx) {}

  protected:
    const FixedRemap &maps_for(int width, int height) override;
    std::unique_ptr<ImageUndistorter> image_undistorter;
};

class OutputWriter {
  public:
    virtual ~OutputWriter() = default;
    virtual void write_pose(const double &t, const OutputPose &pose) = 0;
};
class TumOutputWriter : public OutputWriter {
  public:
    explicit TumOutputWriter(const std::string &filename);
    void write_pose(const double &t, const OutputPose &pose) override;
    bool is_open() const { return file.is_open(); }

  private:
    std::ofstream file;
};

} // namespace pvio
