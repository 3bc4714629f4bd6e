// -*- c++ -*-
// SHADOWS the reference's src/rootba/bal/bal_problem_io.hpp (+ util/serialization.hpp) for the
// oracle/_ref build: load_rootba / save_rootba report failure (see cereal/archives/binary.hpp here).
// TEST INFRASTRUCTURE ONLY.
#pragma once
#include <memory>
#include <string>

#include <cereal/archives/binary.hpp>
#include <glog/logging.h>

#include "rootba/bal/bal_problem.hpp"

namespace rootba {
struct FileInfo {
  std::string type;
  std::string version;
};
static const auto BAL_PROBLEM_FILE_INFO = FileInfo{"rootba::BalProblem", "1.0"};

template <class OutputArchive>
class FileSaver {
 public:
  virtual ~FileSaver() = default;
  virtual bool save() {
    LOG(ERROR) << "oracle/_ref: the .cereal problem cache is not part of this build (" << path_ << ")";
    return false;
  }

 protected:
  explicit FileSaver(const FileInfo& info, std::string path) : info_(info), path_(std::move(path)) {}
  virtual std::string format_summary() const { return ""; }
  virtual bool save_impl(OutputArchive& archive) = 0;
  FileInfo info_;
  std::string path_;
};
template <class InputArchive>
class FileLoader {
 public:
  virtual ~FileLoader() = default;
  virtual bool load() {
    LOG(ERROR) << "oracle/_ref: the .cereal problem cache is not part of this build (" << path_ << ")";
    return false;
  }

 protected:
  explicit FileLoader(const FileInfo& info, std::string path) : info_(info), path_(std::move(path)) {}
  virtual std::string format_summary() const { return ""; }
  virtual bool load_impl() = 0;
  FileInfo info_;
  std::string path_;
  std::unique_ptr<InputArchive> archive_;
};
}  // namespace rootba
