"""CPU oracle (fp64, numpy) for the AcinoSet triangulation + FTE hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``acinoset_amd/`` imports this package;
only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may use it, and only as the checker / the timed CPU baseline.

Each function cites the reference file:line it restates (paths relative to the
upstream AcinoSet tree).  Pinning status:

* ``loss.redescending_loss``, ``camera.pt3d_to_2d``, rotation helpers, cheetah FK
  (positions AND Jacobian), adjacent-pair index path: pinned against vectors
  generated from the reference's own Python (tests/golden/make_golden.py).
* fisheye undistort / 2-view DLT / fisheye projection (OpenCV algorithms, cv2 is
  an un-vendored, unpinned dependency of the reference): pinned jointly by KAT-1,
  the residual statistics recorded in src/calib_with_gui.ipynb cell 29, on the
  shipped sunday_amelia fixtures.
* pinhole ``project_points`` / ``undistort_points`` / ``triangulate_points``
  (cv2.projectPoints / cv2.undistortPoints): PARITY UNPINNED - no recorded
  output in the reference exercises them; restated from OpenCV's documented model.
* the FTE solve (Pyomo + IPOPT, absent): PARITY UNPINNED end to end - no cheetah
  IPOPT trajectory is shipped.  Pinned instead: constants, weights, the 21 boxes,
  the initialisation and the OBJECTIVE by the reference's own model text run on
  floats (fte_model.npz); its GRADIENT and the first-order optimality of the LM
  end point by central differences of that same text (fte_stationary.npz); the
  reduction by KAT-4 (integration / third-difference identities on the stored
  build.py runs).  I.e. the LM solution is a stationary point of the reference's
  NLP; that IPOPT ends at the same one is not shown.  A third-party quasi-Newton
  optimiser (scipy L-BFGS-B) on the same objective ends beside it (fte_lbfgs.npz).
* ``pyomo_model``: this repo's own Pyomo formulation of that NLP (for bench.py's
  conditional IPOPT timing where Pyomo exists); validated against the reference's
  model text through the float stand-ins of tests/golden/_float_pyomo.py.
"""
