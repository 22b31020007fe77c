"""Physical constants and unit conversions used on the hot path.

Values are the reference's (constants.py:29,32,34,44) and must stay bit-identical:
lengths in m, density in g/cm^3, gz in mGal.
"""
#: gravitational constant for density in g/cm^3 (constants.py:34) -- used by prism AND tesseroid gz
G = 0.00000006673
#: SI gravitational constant (constants.py:32) -- not used by gz
Gs = 0.00000000006673
#: m/s^2 -> mGal (constants.py:29)
SI2MGAL = 100000.0
#: mean Earth radius in m (constants.py:44)
MEAN_EARTH_RADIUS = 6378137.0
