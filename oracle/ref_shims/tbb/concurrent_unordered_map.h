// -*- c++ -*-
#pragma once
#include "blocked_range.h"
