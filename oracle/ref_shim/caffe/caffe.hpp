// Stand-in for <caffe/caffe.hpp> -- TEST INFRASTRUCTURE ONLY (see ../README.md).  RetinaFace.h:8,13,80 needs the
// names `caffe`, `Net<float>` and `boost::shared_ptr` to exist; nothing of Caffe is called in the -DUSE_TENSORRT build.
#pragma once
#include <memory>
namespace boost { template <class T> using shared_ptr = std::shared_ptr<T>; }
namespace caffe { template <class T> class Net; }
