"""TEST stand-in"""
