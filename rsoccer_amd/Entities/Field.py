from rsoccer_amd.Entities.records import Field  # noqa: F401
