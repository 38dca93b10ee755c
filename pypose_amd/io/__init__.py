from .g2o import read_g2o, write_g2o, G2OPGO
