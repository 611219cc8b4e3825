"""shared by the k-means fit tests: name -> (init key, n_redo, max_iter, tol key or value, seed key)"""
CASES = {
    "1": ("init", 1, 1, 0.0, None),
    "3": ("init", 1, 3, 0.0, None),
    "tol": ("init", 1, 12, "tol_exit", None),
    "redo": ("init", 2, 3, 0.0, "redo_seed"),
    "redo_b": ("bad_init", 2, 3, 0.0, "redo_b_seed"),
}
