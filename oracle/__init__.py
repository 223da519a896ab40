"""CPU oracle package -- TEST INFRASTRUCTURE ONLY.

Only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` may
import this package.  The product (``weather_model_amd``) never does: its ops raise if the HIP
library is missing.

* ``oracle.cpu``       -- ctypes wrapper over ``wm_oracle.c`` (float64 direct-loop conv2d
                          fwd / bwd-data / bwd-weight+bias, Gram, normal-equation solve).
* ``oracle.model_ref`` -- numpy float64 restatement of ``newConvolution.CustomModel``
                          (forward, backward, MSE, Adam), reference newConvolution.py:109-186.
* ``oracle.generators`` -- numpy restatement of ``WeatherDataGenerator`` (rows, cubes, region
                          walk), reference multipleRegression.py:217-270; arbitrates row a9.
"""
