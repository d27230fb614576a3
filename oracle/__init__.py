"""CPU oracle for the RetinaFace::detect() hot path -- TEST INFRASTRUCTURE ONLY.

This package is a CPU restatement (numpy / PyTorch-CPU fp32 / plain C) of what the
reference computes on the path BASELINE.json names:

  preprocess   reference retinaface/RetinaFace.cpp:950-981   (Caffe variant: pad to x32, BGR->RGB, raw 0..255)
  forward      BVLC Caffe Net::Forward() over model/*.prototxt + model/*.caffemodel
               (reference call site retinaface/RetinaFace.cpp:988; Caffe itself is an
               un-vendored, un-pinned third-party dependency -- CMakeLists.txt:91,120,124)
  decode       reference retinaface/RetinaFace.cpp:999-1072 (+ helpers :9-199, :378-432)
  NMS          reference retinaface/RetinaFace.cpp:434-492

PARITY STATUS: *parity unpinned by the reference*.  The reference ships no tests, no golden
vectors and no runnable build of its own path in this environment (Caffe / TensorRT / OpenCV /
NPP are all absent).  The oracle is therefore pinned by
  (1) two independent forward implementations (PyTorch-CPU conv kernels vs. a plain numpy
      im2col/einsum implementation) that must agree to fp32 round-off,
  (2) a literal numpy restatement of decode/NMS cross-checked against a plain-C restatement,
  (3) the semantic check that data/img.jpg yields the 6 faces at the scores SURVEY.md records.
Golden vectors frozen from it live in tests/golden/ (generator: tools/make_golden.py).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this package.
The product (retinaface_amd/) never does and fails loudly if its HIP library is missing.
"""
