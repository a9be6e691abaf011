// infer_core_amd.hpp -- the part of deploy_core's `inference_core` interface that detection_6d_foundationpose drives
// (zz990099/foundationpose_cpp, D6F/src/foundationpose.cpp:126-139 GetBuffer + GetTensor, :331-354 SetBufferLocation +
// RawPtr + SetShape, :364-365,436 Cast<float>, :207,219,255 SyncInfer; factory call shape
// simple_tests/src/test_foundationpose.cpp:24-35), backed by the MI355X networks through the C ABI (fp_net_*).
// With it the reference's own orchestration code can call the MI355X refiner / scorer unchanged; a caller that builds
// its own BaseInferCore keeps working too, because detection_6d::CreateFoundationPoseModel only asks a core for its
// weights path.  No Eigen / OpenCV needed.
#pragma once

#include <cstdint>
#include <map>
#include <memory>
#include <stdexcept>
#include <string>
#include <unordered_map>
#include <vector>

#include "foundationpose_amd.h"

namespace inference_core {

enum class DataLocation { HOST = 0, DEVICE = 1, UNKOWN = 2 };  // (the reference spells it UNKOWN)

class ITensor {
public:
  virtual ~ITensor() = default;
  virtual const std::string &GetName() const noexcept = 0;
  virtual void *RawPtr() = 0;  // pointer of the copy selected by SetBufferLocation
  template <typename T> T *Cast() { return reinterpret_cast<T *>(RawPtr()); }
  virtual void SetBufferLocation(DataLocation loc) = 0;
  virtual DataLocation GetBufferLocation() const noexcept = 0;
  virtual void SetShape(const std::vector<uint64_t> &shape) = 0;
  virtual const std::vector<uint64_t> &GetShape() const noexcept = 0;
  virtual const std::vector<uint64_t> &GetDefaultShape() const noexcept = 0;
};

class BlobsTensor {
public:
  virtual ~BlobsTensor() = default;
  // throws std::invalid_argument for an unknown name (the reference relies on that to validate engines, foundationpose.cpp:128-139)
  virtual ITensor *GetTensor(const std::string &name) = 0;
};

class BaseInferCore {
public:
  virtual ~BaseInferCore() = default;
  virtual std::shared_ptr<BlobsTensor> GetBuffer(bool block) = 0;
  virtual bool SyncInfer(BlobsTensor *buffer) = 0;
  virtual std::string GetCoreName() const { return "amd-mi355x-core"; }
  // what detection_6d::CreateFoundationPoseModel needs from a core on this platform
  virtual const std::string &WeightsPath() const noexcept = 0;
};

namespace amd_detail {

class Tensor final : public ITensor {
public:
  Tensor(fp_net *net, std::string name, std::vector<uint64_t> shape, bool is_output)
      : net_(net), name_(std::move(name)), shape_(shape), default_shape_(std::move(shape)),
        loc_(is_output ? DataLocation::HOST : DataLocation::DEVICE) {}
  const std::string &GetName() const noexcept override { return name_; }
  void *RawPtr() override {
    void *p = fp_net_blob(net_, name_.c_str(), loc_ == DataLocation::DEVICE ? FP_DEVICE : FP_HOST);
    if (!p) throw std::runtime_error(fp_last_error());
    return p;
  }
  void SetBufferLocation(DataLocation loc) override { loc_ = loc; }
  DataLocation GetBufferLocation() const noexcept override { return loc_; }
  void SetShape(const std::vector<uint64_t> &shape) override {
    if (shape.size() != default_shape_.size() || shape.empty() || shape[0] == 0 || shape[0] > default_shape_[0])
      throw std::invalid_argument("[AmdInferCore] SetShape: batch of '" + name_ + "' out of range");
    for (size_t i = 1; i < shape.size(); i++)
      if (shape[i] != default_shape_[i]) throw std::invalid_argument("[AmdInferCore] SetShape: only the batch dimension of '" + name_ + "' is dynamic");
    shape_ = shape;
  }
  const std::vector<uint64_t> &GetShape() const noexcept override { return shape_; }
  const std::vector<uint64_t> &GetDefaultShape() const noexcept override { return default_shape_; }

private:
  fp_net *net_;
  std::string name_;
  std::vector<uint64_t> shape_, default_shape_;
  DataLocation loc_;
};

class Blobs final : public BlobsTensor {
public:
  ITensor *GetTensor(const std::string &name) override {
    auto it = tensors_.find(name);
    if (it == tensors_.end()) throw std::invalid_argument("[AmdInferCore] no blob named '" + name + "'");
    return it->second.get();
  }
  std::map<std::string, std::unique_ptr<Tensor>> tensors_;
  std::vector<std::string> inputs_, outputs_;
};

}  // namespace amd_detail

// One refiner or scorer network of the MI355X library behind the BaseInferCore contract.  The blob set is fixed by the
// network ("render_input" / "transf_input" -> "trans" + "rot", or -> "scores"); the shape maps of the factory call are
// checked against it, their batch size sizes the buffers.
class AmdInferCore final : public BaseInferCore {
public:
  using ShapeMap = std::unordered_map<std::string, std::vector<uint64_t>>;
  AmdInferCore(const std::string &packed_weights_path, const ShapeMap &inputs, const ShapeMap &outputs) : path_(packed_weights_path) {
    const bool scorer = outputs.count("scores") != 0;
    if (inputs.count("render_input") == 0 || inputs.count("transf_input") == 0 ||
        (scorer ? outputs.size() != 1 : (outputs.count("trans") == 0 || outputs.count("rot") == 0)))
      throw std::invalid_argument("[AmdInferCore] blobs must be render_input + transf_input -> trans + rot (refiner) or -> scores (scorer)");
    const auto &rs = inputs.at("render_input");
    if (rs.size() != 4 || rs[1] != FP_CROP || rs[2] != FP_CROP || rs[3] != 6 || rs != inputs.at("transf_input"))
      throw std::invalid_argument("[AmdInferCore] inputs must be [batch,160,160,6]");
    net_ = fp_net_create(path_.c_str(), scorer ? 1 : 0, (int)rs[0]);
    if (!net_) throw std::runtime_error(fp_last_error());
    blobs_ = std::make_shared<amd_detail::Blobs>();
    for (const auto &kv : inputs) {
      blobs_->tensors_[kv.first] = std::make_unique<amd_detail::Tensor>(net_, kv.first, kv.second, false);
      blobs_->inputs_.push_back(kv.first);
    }
    for (const auto &kv : outputs) {
      blobs_->tensors_[kv.first] = std::make_unique<amd_detail::Tensor>(net_, kv.first, kv.second, true);
      blobs_->outputs_.push_back(kv.first);
    }
  }
  ~AmdInferCore() override { fp_net_destroy(net_); }
  AmdInferCore(const AmdInferCore &) = delete;
  AmdInferCore &operator=(const AmdInferCore &) = delete;

  // mem_buf_size = 1 in the reference's calls (test_foundationpose.cpp:30,35): one buffer set, handed out every time
  std::shared_ptr<BlobsTensor> GetBuffer(bool /*block*/) override { return blobs_; }

  bool SyncInfer(BlobsTensor *buffer) override {
    if (buffer != blobs_.get()) return false;
    auto *render = blobs_->GetTensor("render_input"), *transf = blobs_->GetTensor("transf_input");
    const uint64_t n = render->GetShape()[0];
    if (transf->GetShape()[0] != n) return false;
    int out_loc = FP_DEVICE;
    for (const auto &o : blobs_->outputs_)
      if (blobs_->GetTensor(o)->GetBufferLocation() != DataLocation::DEVICE) out_loc = FP_HOST;
    auto loc = [](ITensor *t) { return t->GetBufferLocation() == DataLocation::DEVICE ? FP_DEVICE : FP_HOST; };
    return fp_net_infer(net_, (int)n, loc(render), loc(transf), out_loc) == 0;
  }
  const std::string &WeightsPath() const noexcept override { return path_; }

private:
  std::string path_;
  fp_net *net_ = nullptr;
  std::shared_ptr<amd_detail::Blobs> blobs_;
};

// same call shape as CreateTrtInferCore(path, {name -> shape}, {name -> shape}, mem_buf_size) (test_foundationpose.cpp:24-35)
inline std::shared_ptr<BaseInferCore> CreateAmdInferCore(const std::string &packed_weights_path, const AmdInferCore::ShapeMap &inputs,
                                                         const AmdInferCore::ShapeMap &outputs, int /*mem_buf_size*/ = 1) {
  return std::make_shared<AmdInferCore>(packed_weights_path, inputs, outputs);
}

// a core that only names the weights file: enough for detection_6d::CreateFoundationPoseModel, which runs the networks itself
class WeightsOnlyCore final : public BaseInferCore {
public:
  explicit WeightsOnlyCore(std::string path) : path_(std::move(path)) {}
  std::shared_ptr<BlobsTensor> GetBuffer(bool) override { throw std::runtime_error("[AmdInferCore] this core carries a weights path only: create it with blob shapes to run it on its own"); }
  bool SyncInfer(BlobsTensor *) override { return false; }
  const std::string &WeightsPath() const noexcept override { return path_; }

private:
  std::string path_;
};
inline std::shared_ptr<BaseInferCore> CreateAmdInferCore(const std::string &packed_weights_path) {
  return std::make_shared<WeightsOnlyCore>(packed_weights_path);
}

}  // namespace inference_core
