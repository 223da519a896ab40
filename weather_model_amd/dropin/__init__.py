"""Same-named mirrors of the reference's Python modules for the hot path.

Either ``from weather_model_amd.dropin import newConvolution`` or put this directory on
``sys.path`` and ``import newConvolution`` exactly as a reference user would.
"""
