"""CPU oracle for the RetinaFace::detect() hot path -- TEST INFRASTRUCTURE ONLY.

This package is a CPU restatement (numpy / PyTorch-CPU fp32 / plain C) of what the
reference computes on the path BASELINE.json names:

  preprocess   reference retinaface/RetinaFace.cpp:950-981   (Caffe variant: pad to x32, BGR->RGB, raw 0..255)
  forward      BVLC Caffe Net::Forward() over model/*.prototxt + model/*.caffemodel
               (reference call site retinaface/RetinaFace.cpp:988; Caffe itself is an
               un-vendored, un-pinned third-party dependency -- CMakeLists.txt:91,120,124)
  decode       reference retinaface/RetinaFace.cpp:999-1072 (+ helpers :9-199, :378-432)
  NMS          reference retinaface/RetinaFace.cpp:434-492

PARITY STATUS: *pinned by the reference's own code* for preprocess / anchors / decode / NMS; the forward is a restatement
*pinned on the reference's MXNet original of the same network* (the only other definition of it the reference holds).
  * The reference's retinaface/RetinaFace.cpp compiles here, unmodified and from where it lies, against stand-in
    third-party headers (oracle/build_ref.py, oracle/ref_shim/, oracle/ref_harness.cpp -> oracle/_ref/).
    tests/test_reference_pin.py holds retinaface_post.py and csrc/rf_post_ref.c bit-exact to that build, live and through
    tests/golden/ref_pin.npz (minted from it by tools/make_ref_golden.py).
  * The forward's arithmetic lives in BVLC Caffe / TensorRT 5.1 -- un-vendored, un-pinned, absent.  caffe_forward.py
    restates Caffe's published layer semantics and is pinned by two independent conv back-ends agreeing to fp32
    round-off plus the semantic check that data/img.jpg yields the 6 faces at the scores SURVEY.md records -- and, since
    round 4, by mxnet_forward.py: an independent interpreter (own reader, own graph walk, own convolution, MXNet operator
    semantics) of MXNet2Caffe/model_mxnet/mnet.25-symbol.json + mnet.25-0000.params, the network model/mnet25.caffemodel was
    converted from.  tests/test_mxnet_pin.py: all 273 parameter arrays bit-equal under MXNet2Caffe/mxnet2caffe.py:42-113;
    on the all-ones tensor of MXNet2Caffe/check_results.py:29 and on a crop of data/img.jpg the stride-32 heads agree to
    <= 1e-5 of range, strides 16 / 8 to <= 3e-5 once the Caffe graph's bilinear deconvolution is swapped for the MXNet
    graph's nearest x2, and every one of the 156 intermediate nodes agrees too (goldens: tests/golden/mxnet_pin.npz,
    minted by tools/make_mxnet_golden.py).
  * NPP's closed-source SUPER resize (oversize frames) stays "parity unpinned".
Golden vectors live in tests/golden/ (generators: tools/make_ref_golden.py, tools/make_golden.py).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this package.
The product (retinaface_amd/) never does and fails loudly if its HIP library is missing.
"""
