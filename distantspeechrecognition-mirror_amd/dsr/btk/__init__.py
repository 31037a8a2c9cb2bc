"""Python face of the BTK operators with the reference's SWIG-visible names (btk/*/*.i): every class X is exposed
as XPtr(...), `for v in stream` calls reset() then next() until the end-of-stream error becomes StopIteration
(btk/stream/stream.i:111-115, btk/include/jexception.i:178-180)."""
