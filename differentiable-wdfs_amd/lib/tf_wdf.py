"""Drop-in for wdf_py/lib/tf_wdf.py (placeholder: the element API lands next)."""
