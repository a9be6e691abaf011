// foundationpose_amd.hpp -- dependency-free C++17 RAII wrapper over the C ABI (foundationpose_amd.h).
// Mirrors detection_6d::Base6DofDetectionModel (reference detection_6d_foundationpose/include/
// detection_6d_foundationpose/foundationpose.hpp:16-77) with plain structs instead of cv::Mat / Eigen types, so it
// builds where OpenCV and Eigen do not exist (the MI355X image).  detection_6d_foundationpose_amd.hpp layers the
// reference's exact signatures on top when those headers are present.
#pragma once

#include <array>
#include <cstdint>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

#include "foundationpose_amd.h"

namespace fp_amd {

using Pose = std::array<float, 16>;  // column-major 4x4 (Eigen::Matrix4f::data())

struct ImageU8 { const uint8_t *data; int rows, cols, channels; };
struct ImageF32 { const float *data; int rows, cols; };

struct Mesh {  // what BaseMeshLoader's getters return (mesh_loader.hpp:25-61)
  std::string name;
  std::vector<float> vertices, normals, texcoords;  // [V,3], [V,3], [V,2]
  std::vector<uint32_t> faces;                      // [F,3]
  std::vector<uint8_t> texture;                     // [TH,TW,3] RGB
  int tex_height = 0, tex_width = 0;
  float diameter = 0;
  float center[3] = {0, 0, 0};
};

// CreateAssimpMeshLoader(name, path) equivalent (mesh_loader.hpp:92-93): OBJ + MTL + PNG; throws like the reference
inline Mesh LoadObjMesh(const std::string &name, const std::string &mesh_file_path, float *orient_bounds16 = nullptr,
                        float *dimension3 = nullptr) {
  fp_loaded_mesh *h = fp_mesh_load_obj(name.c_str(), mesh_file_path.c_str());
  if (!h) throw std::runtime_error(fp_last_error());
  const fp_mesh *v = fp_mesh_view(h);
  Mesh m;
  m.name = name;
  m.vertices.assign(v->vertices, v->vertices + (size_t)v->num_vertices * 3);
  m.normals.assign(v->normals, v->normals + (size_t)v->num_vertices * 3);
  m.texcoords.assign(v->texcoords, v->texcoords + (size_t)v->num_vertices * 2);
  m.faces.assign(v->faces, v->faces + (size_t)v->num_faces * 3);
  m.texture.assign(v->texture, v->texture + (size_t)v->tex_height * v->tex_width * 3);
  m.tex_height = v->tex_height; m.tex_width = v->tex_width; m.diameter = v->diameter;
  for (int k = 0; k < 3; k++) m.center[k] = v->center[k];
  fp_mesh_orient_bounds(h, orient_bounds16, dimension3);
  fp_mesh_free(h);
  return m;
}

class FoundationPose {
public:
  // CreateFoundationPoseModel (foundationpose.hpp:99-105); throws std::runtime_error like the reference constructor
  FoundationPose(const std::vector<Mesh> &meshes, const float K[9], const std::string &refiner_weights,
                 const std::string &scorer_weights, int max_h = 1080, int max_w = 1920) {
    std::vector<fp_mesh> cm(meshes.size());
    for (size_t i = 0; i < meshes.size(); i++) {
      const Mesh &m = meshes[i];
      cm[i].name = m.name.c_str();
      cm[i].num_vertices = (int)(m.vertices.size() / 3);
      cm[i].num_faces = (int)(m.faces.size() / 3);
      cm[i].vertices = m.vertices.data(); cm[i].normals = m.normals.data(); cm[i].texcoords = m.texcoords.data();
      cm[i].faces = m.faces.data(); cm[i].texture = m.texture.data();
      cm[i].tex_height = m.tex_height; cm[i].tex_width = m.tex_width;
      cm[i].diameter = m.diameter;
      for (int k = 0; k < 3; k++) cm[i].center[k] = m.center[k];
    }
    h_ = fp_create(cm.data(), (int)cm.size(), K, refiner_weights.empty() ? nullptr : refiner_weights.c_str(),
                   scorer_weights.empty() ? nullptr : scorer_weights.c_str(), max_h, max_w);
    if (!h_) throw std::runtime_error(std::string("[FoundationPose] Failed to Construct FoundationPose, ex : ") + fp_last_error());
  }
  ~FoundationPose() { fp_destroy(h_); }
  FoundationPose(const FoundationPose &) = delete;
  FoundationPose &operator=(const FoundationPose &) = delete;

  // Register (foundationpose.hpp:36-41): false on failure, message in last_error()
  bool Register(const ImageU8 &rgb, const ImageF32 &depth, const ImageU8 &mask, const std::string &target_name,
                Pose &out_pose_in_mesh, size_t refine_itr = 1) {
    if (rgb.rows != depth.rows || rgb.cols != depth.cols || mask.rows != depth.rows || mask.cols != depth.cols) {
      err_ = "[FoundationPose] Got rgb/depth/mask with different size!";
      return false;
    }
    return ok(fp_register(h_, rgb.data, depth.data, mask.data, depth.rows, depth.cols, target_name.c_str(),
                          (int)refine_itr, out_pose_in_mesh.data()));
  }
  // Track (foundationpose.hpp:59-64)
  bool Track(const ImageU8 &rgb, const ImageF32 &depth, const Pose &hyp_pose_in_mesh, const std::string &target_name,
             Pose &out_pose_in_mesh, size_t refine_itr = 1) {
    if (rgb.rows != depth.rows || rgb.cols != depth.cols) {
      err_ = "[FoundationPose] Got rgb/depth/mask with different size!";
      return false;
    }
    return ok(fp_track(h_, rgb.data, depth.data, depth.rows, depth.cols, hyp_pose_in_mesh.data(), target_name.c_str(),
                       (int)refine_itr, out_pose_in_mesh.data()));
  }
  // Track in two halves (pipelined serving: one thread, several objects in flight): the frame must stay valid until TrackWait
  bool TrackSubmit(const ImageU8 &rgb, const ImageF32 &depth, const Pose &hyp_pose_in_mesh, const std::string &target_name,
                   size_t refine_itr = 1) {
    if (rgb.rows != depth.rows || rgb.cols != depth.cols) {
      err_ = "[FoundationPose] Got rgb/depth/mask with different size!";
      return false;
    }
    return ok(fp_track_submit(h_, rgb.data, depth.data, FP_HOST, depth.rows, depth.cols, hyp_pose_in_mesh.data(), target_name.c_str(),
                              (int)refine_itr));
  }
  bool TrackWait(Pose &out_pose_in_mesh) { return ok(fp_track_wait(h_, out_pose_in_mesh.data())); }
  // K objects of one frame as one batch (geometry per object, one refine-net pass over all crops)
  bool TrackMulti(const ImageU8 &rgb, const ImageF32 &depth, const std::vector<Pose> &hyp_poses, const std::vector<std::string> &target_names,
                  std::vector<Pose> &out_poses, size_t refine_itr = 1) {
    if (rgb.rows != depth.rows || rgb.cols != depth.cols || hyp_poses.size() != target_names.size() || hyp_poses.empty()) {
      err_ = "[FoundationPose] TrackMulti: inconsistent arguments";
      return false;
    }
    std::vector<const char *> names;
    for (const auto &n : target_names) names.push_back(n.c_str());
    out_poses.resize(hyp_poses.size());
    return ok(fp_track_multi(h_, rgb.data, depth.data, FP_HOST, depth.rows, depth.cols, (int)hyp_poses.size(), hyp_poses[0].data(), names.data(),
                             (int)refine_itr, out_poses[0].data()));
  }

  // ---- options the reference does not have ----
  // element type of both networks: FP_PREC_F16 (default, the reference's TensorRT --fp16), FP_PREC_BF16, FP_PREC_FP8 / FP_PREC_INT8 (after Calibrate)
  bool SetPrecision(int precision) { return ok(fp_set_precision(h_, precision)); }
  int precision() const { return fp_get_precision(h_); }
  // one f16 Register of a representative frame that records the per-activation maxima the static FP8 quantisation needs
  bool CalibrateFp8(const ImageU8 &rgb, const ImageF32 &depth, const ImageU8 &mask, const std::string &target_name) {
    return ok(fp_calibrate_fp8(h_, rgb.data, depth.data, mask.data, FP_HOST, depth.rows, depth.cols, target_name.c_str()));
  }
  // post-training quantisation of an 8-bit precision on a representative frame (scales + bias correction); the record can be saved and restored
  bool Calibrate(const ImageU8 &rgb, const ImageF32 &depth, const ImageU8 &mask, const std::string &target_name, int precision) {
    return ok(fp_calibrate(h_, rgb.data, depth.data, mask.data, FP_HOST, depth.rows, depth.cols, target_name.c_str(), precision));
  }
  // ... over several frames of the deployment's scene family (K >= 8 recommended: the common-mode correction then carries over to unseen frames)
  bool CalibrateBegin(int precision) { return ok(fp_calibrate_begin(h_, precision)); }
  bool CalibrateAddFrame(const ImageU8 &rgb, const ImageF32 &depth, const ImageU8 &mask, const std::string &target_name) {
    return ok(fp_calibrate_add_frame(h_, rgb.data, depth.data, mask.data, FP_HOST, depth.rows, depth.cols, target_name.c_str()));
  }
  bool CalibrateFinish() { return ok(fp_calibrate_finish(h_)); }
  void CalibrateAbort() { (void)fp_calibrate_abort(h_); }
  bool GetCalibrationBlob(int precision, std::vector<unsigned char> &blob) const {
    blob.resize(fp_calibration_size());
    return fp_get_calibration_blob(h_, precision, blob.data(), blob.size()) == 0;
  }
  bool SetCalibrationBlob(const std::vector<unsigned char> &blob) { return ok(fp_set_calibration_blob(h_, blob.data(), blob.size())); }
  bool GetCalibration(std::array<float, 32> &amax) const { return fp_get_calibration(h_, amax.data()) == 0; }
  bool SetCalibration(const std::array<float, 32> &amax) { return ok(fp_set_calibration(h_, amax.data())); }
  // float model of the rendering stage: FP_FLOAT_FMAD (default: contracted like the reference's nvcc build) or FP_FLOAT_SEPARATE
  bool SetFloatModel(int model) { return ok(fp_set_float_model(h_, model)); }

  const std::string &last_error() const { return err_; }
  fp_model *handle() { return h_; }

private:
  bool ok(int rc) {
    err_ = rc ? fp_last_error() : "";
    return rc == 0;
  }
  fp_model *h_ = nullptr;
  std::string err_;
};

}  // namespace fp_amd
