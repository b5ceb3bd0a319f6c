from rsoccer_amd.Entities.records import Robot  # noqa: F401
