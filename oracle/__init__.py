"""CPU oracle for the sr355 hot path -- TEST INFRASTRUCTURE ONLY.

This package restates, on the CPU (torch-CPU / NumPy, fp32 with an fp64 switch), the
algorithms the reference executes through TensorFlow/Keras/OpenCV on its hot path
(SURVEY.md section 8a).  Each function cites the reference file:line it follows.

PARITY UNPINNED (numerically): the reference's arithmetic lives in un-vendored third-party
dependencies -- TensorFlow 2.10.0 / its bundled Keras (version printed in
SRModels/deep_learning_models/ESRGAN.ipynb:L23), tensorflow-addons, OpenCV, scikit-image
(versions unpinned: the reference has no requirements file) -- none of which is installed
here, the reference has no tests or golden vectors, and no trained weights exist in the
snapshot.  What *is* pinned, and asserted in tests/test_oracle_pins.py, are the structural
known-answers left in the reference notebooks' outputs: per-model parameter counts and
dataset patch counts.  Library semantics (Conv2D SAME, depth_to_space DCR order,
tf.image.psnr/ssim, OpenCV INTER_CUBIC, np.pad reflect) are restated from their published
definitions and cross-checked in tests by independent second
derivations (naive fp64 NumPy loops, torch's own bicubic, scipy filters).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this package.
The product (the sr355 package + libsr355.so) never does.
"""
