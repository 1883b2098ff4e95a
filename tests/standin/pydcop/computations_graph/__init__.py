"""TEST stand-in"""
__path__ = list(__path__)
