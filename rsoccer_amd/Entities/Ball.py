from rsoccer_amd.Entities.records import Ball  # noqa: F401
