"""Drop-in for the reference's `simple_knn` package (imported at scene/gaussian_model.py:20): `simple_knn._C.distCUDA2`."""
