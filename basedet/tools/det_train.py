"""`basedet_train` under the reference's module path (tools/det_train.py): the MI355X trainer entry of basedet_amd."""
from basedet_amd.tools.det_train import default_parser, load_cfg, main, worker  # noqa: F401

if __name__ == "__main__":
    main()
