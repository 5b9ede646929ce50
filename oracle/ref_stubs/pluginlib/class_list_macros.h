// ORACLE (test infrastructure only): the export macro of pluginlib does nothing here
#pragma once
#define PLUGINLIB_EXPORT_CLASS(a, b)
