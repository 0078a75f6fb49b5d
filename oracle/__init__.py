"""CPU checkers for the box pipeline -- TEST INFRASTRUCTURE, never imported by the product.

PARITY UNPINNED (torchvision 0.16.2 absent; the reference holds no fixtures).  See
``oracle/tv_reference.py`` (torch restatement), ``oracle/od_oracle.c`` (independent scalar C
restatement) and ``oracle/spec_math.py`` (bit-reproducible exp / sigmoid).
"""
