// -*- c++ -*-
// Stand-in for basalt-headers' assertion macros (TEST INFRASTRUCTURE ONLY, see Eigen/Dense here).
#pragma once
#include <cstdlib>
#include <iostream>
#define BASALT_ASSERT(expr)                                                                  \
  do {                                                                                       \
    if (!(expr)) {                                                                           \
      std::cerr << "BASALT_ASSERT failed: " #expr " (" << __FILE__ << ":" << __LINE__ << ")" \
                << std::endl;                                                                \
      std::abort();                                                                          \
    }                                                                                        \
  } while (0)
#define BASALT_ASSERT_MSG(expr, msg) BASALT_ASSERT(expr)
#define BASALT_ASSERT_STREAM(expr, msg) BASALT_ASSERT(expr)
