"""TEST stand-in, see tests/standin/README.md"""
STANDIN = True
