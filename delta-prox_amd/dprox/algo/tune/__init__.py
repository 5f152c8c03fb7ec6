from .dpir import log_descent
