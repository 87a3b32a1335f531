"""Finger-geometry decode on the device: the part of the reference's ``assets/`` package that sits directly behind the sampler
(SURVEY.md §8(f) rank 3).  Mesh extrusion, convex decomposition and MuJoCo XML generation stay with the user's simulator setup."""
