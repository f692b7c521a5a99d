from .auxiliary import ppException


class LoadflowNotConverged(ppException):
    """pandapower.powerflow.LoadflowNotConverged"""
