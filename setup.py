"""In-tree package metadata: the `basedet_train` console entry of the reference (setup.py:31-39) points at the MI355X trainer.
The HIP library is built in-tree by `python -m basedet_amd.build` (hipcc, gfx950), not by setuptools."""
import setuptools

setuptools.setup(
    name="basedet_amd",
    version="0.2.0",
    description="MI355X-native training hot path behind BaseDet's operator / model / solver surface",
    packages=setuptools.find_packages(include=["basedet_amd*", "basedet", "basedet.*"]),
    package_data={"basedet_amd": ["lib/*.so"]},
    entry_points={"console_scripts": ["basedet_train=basedet_amd.tools.det_train:main"]},
)
