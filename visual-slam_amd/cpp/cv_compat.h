// Minimal stand-ins for the few OpenCV types that appear in the reference's ORBextractor / ORBmatcher
// signatures (include/ORBextractor.h:102-104, include/ORBmatcher.h).  Only used when the real OpenCV is
// not available (this image and the GPU box have none); define ORBX_WITH_OPENCV to build against the
// real headers instead.  Layouts match OpenCV: KeyPoint is 28 bytes, Mat rows are `step` bytes apart.
#pragma once
#ifdef ORBX_WITH_OPENCV
#include <opencv2/core/core.hpp>
#else
#include "cv_standin.h"
#endif
