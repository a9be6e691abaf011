// detection_6d_foundationpose_amd.hpp -- the reference's own C++ API on top of the MI355X library.
//
// Drop-in for detection_6d_foundationpose/include/detection_6d_foundationpose/foundationpose.hpp:16-105 and
// mesh_loader.hpp:15-93 of zz990099/foundationpose_cpp: same namespace, class and factory names, same argument meaning,
// same error behaviour (bool + message; constructors throw std::runtime_error).  Needs OpenCV + Eigen, which the
// MI355X build image does not have, so it is compiled only by the reference-side build; the
// dependency-free equivalent is foundationpose_amd.hpp.
//
// The reference constructs the model from two `inference_core::BaseInferCore` handles wrapping TensorRT engines
// (foundationpose.cpp:448-458).  infer_core_amd.hpp provides that interface on the MI355X networks: `CreateAmdInferCore(path,
// inputs, outputs)` has CreateTrtInferCore's call shape and is a full core (GetBuffer / GetTensor / SetShape / SyncInfer);
// `CreateAmdInferCore(path)` only names the weights.  simple_tests changes exactly its two factory calls
// (test_foundationpose.cpp:24-35).
#pragma once
#if __has_include(<Eigen/Dense>) && __has_include(<opencv2/core.hpp>)

#include <Eigen/Dense>
#include <cstring>
#include <memory>
#include <opencv2/core.hpp>
#include <stdexcept>
#include <string>
#include <vector>

#include "foundationpose_amd.h"
#include "infer_core_amd.hpp"  // inference_core::BaseInferCore / ITensor / BlobsTensor / CreateAmdInferCore

namespace detection_6d {

class BaseMeshLoader {  // mesh_loader.hpp:15-62, unchanged
public:
  virtual ~BaseMeshLoader() = default;
  using Vector3ui = Eigen::Matrix<uint32_t, 3, 1>;
  virtual std::string GetName() const noexcept = 0;
  virtual float GetMeshDiameter() const noexcept = 0;
  virtual size_t GetMeshNumVertices() const noexcept = 0;
  virtual size_t GetMeshNumFaces() const noexcept = 0;
  virtual const std::vector<Eigen::Vector3f> &GetMeshVertices() const noexcept = 0;
  virtual const std::vector<Eigen::Vector3f> &GetMeshVertexNormals() const noexcept = 0;
  virtual const std::vector<Eigen::Vector3f> &GetMeshTextureCoords() const noexcept = 0;
  virtual const std::vector<Vector3ui> &GetMeshTriangleFaces() const noexcept = 0;
  virtual const Eigen::Vector3f &GetMeshModelCenter() const noexcept = 0;
  virtual const Eigen::Matrix4f &GetOrientBounds() const noexcept = 0;
  virtual const Eigen::Vector3f &GetObjectDimension() const noexcept = 0;
  virtual const cv::Mat &GetTextureMap() const noexcept = 0;
};

inline Eigen::Matrix4f ConvertPoseMesh2BBox(const Eigen::Matrix4f &pose_in_mesh,
                                            const std::shared_ptr<BaseMeshLoader> &mesh_loader) {  // mesh_loader.hpp:75-81
  Eigen::Matrix4f tf_to_center = Eigen::Matrix4f::Identity();
  tf_to_center.block<3, 1>(0, 3) = -mesh_loader->GetMeshModelCenter();
  return pose_in_mesh * tf_to_center * mesh_loader->GetOrientBounds();
}

// CreateAssimpMeshLoader(name, path) (mesh_loader.hpp:92-93) without assimp: Wavefront OBJ + MTL + PNG through
// fp_mesh_load_obj.  Throws std::runtime_error where the reference throws (empty path, unreadable file, no UVs).
class ObjMeshLoader : public BaseMeshLoader {
public:
  ObjMeshLoader(const std::string &name, const std::string &mesh_file_path) : name_(name) {
    fp_loaded_mesh *h = fp_mesh_load_obj(name.c_str(), mesh_file_path.c_str());
    if (!h) throw std::runtime_error(fp_last_error());
    const fp_mesh *m = fp_mesh_view(h);
    for (int i = 0; i < m->num_vertices; i++) {
      vertices_.emplace_back(m->vertices[3 * i], m->vertices[3 * i + 1], m->vertices[3 * i + 2]);
      normals_.emplace_back(m->normals[3 * i], m->normals[3 * i + 1], m->normals[3 * i + 2]);
      uvs_.emplace_back(m->texcoords[2 * i], m->texcoords[2 * i + 1], 0.f);  // aiVector3D-style (u, v, 0)
    }
    for (int i = 0; i < m->num_faces; i++) {
      Vector3ui f;
      f[0] = m->faces[3 * i]; f[1] = m->faces[3 * i + 1]; f[2] = m->faces[3 * i + 2];
      faces_.push_back(f);
    }
    texture_ = cv::Mat(m->tex_height, m->tex_width, CV_8UC3);
    std::memcpy(texture_.data, m->texture, (size_t)m->tex_height * m->tex_width * 3);
    diameter_ = m->diameter;
    center_ = Eigen::Vector3f(m->center[0], m->center[1], m->center[2]);
    float ob[16], dim[3];
    fp_mesh_orient_bounds(h, ob, dim);
    for (int c = 0; c < 4; c++)
      for (int r = 0; r < 4; r++) orient_bounds_(r, c) = ob[c * 4 + r];
    dimension_ = Eigen::Vector3f(dim[0], dim[1], dim[2]);
    fp_mesh_free(h);
  }
  std::string GetName() const noexcept override { return name_; }
  float GetMeshDiameter() const noexcept override { return diameter_; }
  size_t GetMeshNumVertices() const noexcept override { return vertices_.size(); }
  size_t GetMeshNumFaces() const noexcept override { return faces_.size(); }
  const std::vector<Eigen::Vector3f> &GetMeshVertices() const noexcept override { return vertices_; }
  const std::vector<Eigen::Vector3f> &GetMeshVertexNormals() const noexcept override { return normals_; }
  const std::vector<Eigen::Vector3f> &GetMeshTextureCoords() const noexcept override { return uvs_; }
  const std::vector<Vector3ui> &GetMeshTriangleFaces() const noexcept override { return faces_; }
  const Eigen::Vector3f &GetMeshModelCenter() const noexcept override { return center_; }
  const Eigen::Matrix4f &GetOrientBounds() const noexcept override { return orient_bounds_; }
  const Eigen::Vector3f &GetObjectDimension() const noexcept override { return dimension_; }
  const cv::Mat &GetTextureMap() const noexcept override { return texture_; }

private:
  std::string name_;
  float diameter_ = 0;
  std::vector<Eigen::Vector3f> vertices_, normals_, uvs_;
  std::vector<Vector3ui> faces_;
  Eigen::Vector3f center_, dimension_;
  Eigen::Matrix4f orient_bounds_;
  cv::Mat texture_;
};
inline std::shared_ptr<BaseMeshLoader> CreateObjMeshLoader(const std::string &name, const std::string &mesh_file_path) {
  return std::make_shared<ObjMeshLoader>(name, mesh_file_path);
}
// same name as the reference's factory, so simple_tests needs no edit for the mesh either
inline std::shared_ptr<BaseMeshLoader> CreateAssimpMeshLoader(const std::string &name, const std::string &mesh_file_path) {
  return CreateObjMeshLoader(name, mesh_file_path);
}

class Base6DofDetectionModel {  // foundationpose.hpp:16-77, unchanged
public:
  virtual bool Register(const cv::Mat &rgb, const cv::Mat &depth, const cv::Mat &mask, const std::string &target_name,
                        Eigen::Matrix4f &out_pose_in_mesh, size_t refine_itr = 1) = 0;
  virtual bool Track(const cv::Mat &rgb, const cv::Mat &depth, const Eigen::Matrix4f &hyp_pose_in_mesh,
                     const std::string &target_name, Eigen::Matrix4f &out_pose_in_mesh, size_t refine_itr = 1) = 0;
  virtual ~Base6DofDetectionModel() = default;
protected:
  Base6DofDetectionModel() = default;
};

class FoundationPoseAmd : public Base6DofDetectionModel {
public:
  FoundationPoseAmd(std::shared_ptr<inference_core::BaseInferCore> refiner, std::shared_ptr<inference_core::BaseInferCore> scorer,
                    const std::vector<std::shared_ptr<BaseMeshLoader>> &loaders, const Eigen::Matrix3f &K, int max_h, int max_w) {
    std::vector<fp_mesh> cm(loaders.size());
    std::vector<std::vector<float>> v(loaders.size()), n(loaders.size()), uv(loaders.size());
    std::vector<std::vector<uint32_t>> f(loaders.size());
    std::vector<std::string> names(loaders.size());
    std::vector<cv::Mat> tex(loaders.size());
    for (size_t i = 0; i < loaders.size(); i++) {
      const auto &L = *loaders[i];
      names[i] = L.GetName();
      for (const auto &p : L.GetMeshVertices()) v[i].insert(v[i].end(), {p[0], p[1], p[2]});
      for (const auto &p : L.GetMeshVertexNormals()) n[i].insert(n[i].end(), {p[0], p[1], p[2]});
      for (const auto &p : L.GetMeshTextureCoords()) uv[i].insert(uv[i].end(), {p[0], p[1]});
      for (const auto &t : L.GetMeshTriangleFaces()) f[i].insert(f[i].end(), {t[0], t[1], t[2]});
      tex[i] = L.GetTextureMap().isContinuous() ? L.GetTextureMap() : L.GetTextureMap().clone();
      if (tex[i].channels() != 3) throw std::runtime_error("[FoundationPose Renderer] Failed to load textured mesh!!!");
      cm[i] = fp_mesh{names[i].c_str(), (int)L.GetMeshNumVertices(), (int)L.GetMeshNumFaces(), v[i].data(), n[i].data(),
                      uv[i].data(), f[i].data(), tex[i].data, tex[i].rows, tex[i].cols, L.GetMeshDiameter(),
                      {L.GetMeshModelCenter()[0], L.GetMeshModelCenter()[1], L.GetMeshModelCenter()[2]}};
    }
    Eigen::Matrix<float, 3, 3, Eigen::RowMajor> Kr = K;
    h_ = fp_create(cm.data(), (int)cm.size(), Kr.data(), refiner->WeightsPath().c_str(), scorer->WeightsPath().c_str(), max_h, max_w);
    if (!h_) throw std::runtime_error(std::string("[FoundationPose] Failed to Construct FoundationPose, ex : ") + fp_last_error());
  }
  ~FoundationPoseAmd() override { fp_destroy(h_); }

  bool Register(const cv::Mat &rgb, const cv::Mat &depth, const cv::Mat &mask, const std::string &target_name,
                Eigen::Matrix4f &out_pose_in_mesh, size_t refine_itr = 1) override {
    if (rgb.size() != depth.size() || mask.size() != depth.size()) return false;  // CheckInputArguments
    cv::Mat r = rgb.isContinuous() ? rgb : rgb.clone(), d = depth.isContinuous() ? depth : depth.clone(),
            m = mask.isContinuous() ? mask : mask.clone();
    return fp_register(h_, r.data, reinterpret_cast<const float *>(d.data), m.data, d.rows, d.cols, target_name.c_str(),
                       (int)refine_itr, out_pose_in_mesh.data()) == 0;  // Eigen::Matrix4f is column-major
  }
  bool Track(const cv::Mat &rgb, const cv::Mat &depth, const Eigen::Matrix4f &hyp_pose_in_mesh, const std::string &target_name,
             Eigen::Matrix4f &out_pose_in_mesh, size_t refine_itr = 1) override {
    if (rgb.size() != depth.size()) return false;
    cv::Mat r = rgb.isContinuous() ? rgb : rgb.clone(), d = depth.isContinuous() ? depth : depth.clone();
    return fp_track(h_, r.data, reinterpret_cast<const float *>(d.data), d.rows, d.cols, hyp_pose_in_mesh.data(),
                    target_name.c_str(), (int)refine_itr, out_pose_in_mesh.data()) == 0;
  }

  // the C-ABI handle: options the reference does not have (fp_set_precision, fp_calibrate_fp8, fp_set_float_model, the hypothesis-
  // shard entry points ...) are reached through it: `auto *amd = dynamic_cast<detection_6d::FoundationPoseAmd *>(model.get());`
  fp_model *handle() { return h_; }

private:
  fp_model *h_ = nullptr;
};

inline std::shared_ptr<Base6DofDetectionModel> CreateFoundationPoseModel(  // foundationpose.hpp:99-105
    std::shared_ptr<inference_core::BaseInferCore> refiner_core, std::shared_ptr<inference_core::BaseInferCore> scorer_core,
    const std::vector<std::shared_ptr<BaseMeshLoader>> &mesh_loaders, const Eigen::Matrix3f &intrinsic_in_mat,
    const int max_input_image_height = 1080, const int max_input_image_width = 1920) {
  return std::make_shared<FoundationPoseAmd>(refiner_core, scorer_core, mesh_loaders, intrinsic_in_mat,
                                             max_input_image_height, max_input_image_width);
}

}  // namespace detection_6d
#endif
