"""Drop-in for the reference's `simple_knn` package (scene/gaussian_model.py:22)."""
