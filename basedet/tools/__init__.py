"""`basedet.tools`: the command-line entries (real modules, so that `python -m basedet.tools.det_train` resolves)."""
