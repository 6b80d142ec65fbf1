"""Small host-side utilities of the training driver (schedules, bookkeeping, timing)."""
