// Compile-only stand-in: cv::KeyPoint comes with the core stand-in.
#pragma once
#include "../core/core.hpp"
