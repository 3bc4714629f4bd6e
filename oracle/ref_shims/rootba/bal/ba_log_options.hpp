// -*- c++ -*-
// SHADOWS the reference's src/rootba/bal/ba_log_options.hpp for the oracle/_ref build (TEST
// INFRASTRUCTURE ONLY): the log-file options are a member of SolverOptions but play no part in the
// solver; the reference's header pulls in the JSON log structures and the flags library.
#pragma once
#include <string>
namespace rootba {
struct BaLogOptions {
  std::string log_path = "ba_log.json";
  int save_log_flags = 1;
  bool disable_all = false;
};
}  // namespace rootba
