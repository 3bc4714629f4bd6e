// -*- c++ -*-
#pragma once
#include "format.h"
