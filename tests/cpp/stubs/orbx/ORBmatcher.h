#pragma once
#include "../../../../visual-slam_amd/cpp/ORBmatcher.h"
